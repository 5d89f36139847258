import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import measure_cell_contraction
print(json.dumps(measure_cell_contraction(M.Context(), torch, 256), indent=1))
