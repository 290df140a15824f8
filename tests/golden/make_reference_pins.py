"""Generates tests/golden/reference_pins.json from the reference tree (run in the build
container only; /root/reference does not exist on the GPU box).

Two structural pins the reference itself holds for the hot path:
  * the 64 adapter 8-mer hash indices hard-coded in bamqc_summary.py (adapter_8_mers), which
    fix the orientation of OverallNumbers::count8mers' index;
  * bamqc_summary.triplet_seq(i) for i in 0..63, which fixes contextToIndex's digit order.
The first is read from the file as data (a list of integers); the second is produced by
importing the reference's Python module and calling its function.
"""
import importlib.util
import json
import os
import re
import sys

REF = os.environ.get("BAMQC_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "bamqc_summary.py")).read()
m = re.search(r"def adapter_8_mers\(kmer_counts\):.*?indices = \[([0-9,\s]+)\]", src, re.S)
indices = [int(x) for x in m.group(1).split(",")]
spec = importlib.util.spec_from_file_location("bamqc_summary_ref", os.path.join(REF, "bamqc_summary.py"))
mod = importlib.util.module_from_spec(spec)
sys.argv = ["bamqc_summary.py"]
spec.loader.exec_module(mod)
triplets = [mod.triplet_seq(i) for i in range(64)]
out = {"source": "DecodeGenetics/BamQC bamqc_summary.py (adapter_8_mers indices; triplet_seq)",
       "adapter_8mer_indices": indices, "triplet_seq": triplets}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "reference_pins.json"), "w"), indent=1)
print(len(indices), triplets[:4], triplets[27])
