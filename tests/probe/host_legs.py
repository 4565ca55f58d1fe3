"""The two host-bound legs of bench.py alone (unchanged spec; one rank through the distributed path), without the rest."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import bench
a = bench.parse_args([]) if hasattr(bench, "parse_args") else None
