"""harness/ -- bench / test harnesses, NOT part of the product package `link_amd` (nothing under link_amd/ imports this):

networks.py   the reference's segmentation networks (ELKUNet / ELKEncoder shapes, linkunet.py / linkencoder.py) assembled from
              link_amd modules with the reference's attribute names, so that reference checkpoints load strict=True -- what
              bench.py --workload cfg3 / cfg4 and the network-level parity tests run
bevhead.py    plain-torch stand-in for the dense BEV half of BASELINE.json configs[4] (RPN + CenterHead.forward), used only by
              bench.py --workload cfg5 --bev
"""
