import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3sc_amd.engine import BellmanEngine
e = BellmanEngine(0)
print("blocks/CU", os.environ.get("C3SC_PEAK_BLOCKS_PER_CU", "8"), "fma f64 TF", round(e.peak_fma_f64(), 2), "mfma f64 TF", round(e.peak_mfma_f64(), 2))
