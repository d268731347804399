"""Offset-mode AGCL (channels-last kernel): pixels per block vs map size.   NND_AGCL_PB=8|16|64 python scripts/sweep_agcl.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nndepth_amd import profiling
print("NND_AGCL_PB =", os.environ.get("NND_AGCL_PB", "(auto)"))
print(profiling.format_rows([r for r in profiling.cre_rows("cuda:0") if "offset" in r["kernel"]]))
