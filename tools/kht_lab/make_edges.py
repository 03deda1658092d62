"""Writes the Canny(59,119) edge map of the 4K benchmark frame (CPU oracle) as raw bytes: input of link_bench."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_bindings import Oracle, synth_frame
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
rc, edges = Oracle().canny(synth_frame(W, H, 12345), 59.0, 119.0)
edges.tofile(sys.argv[1])
