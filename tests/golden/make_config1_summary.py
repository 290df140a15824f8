"""Generates tests/golden/config1_summary.json: the reference's downstream consumer (bamqc_summary.py: read_bamqc_output +
summarize) applied to the oracle's `.bamqc` of the seeded config-1 reads reduced to complete pairs (see tests/test_summary_consumer.py).  Run in the build container only (/root/reference does not exist on the
GPU box).  The fixture pins that files in our output format stay parseable by the reference's summary tool and what it
derives from them."""
import importlib.util
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("BAMQC_REFERENCE", "/root/reference")


def load_reference_module():
    spec = importlib.util.spec_from_file_location("bamqc_summary_ref", os.path.join(REF, "bamqc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bamqc_summary.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def summarize_file(mod, path):
    data = []
    mod.read_bamqc_output(data, path)
    out = []
    for lane in data:
        summary = {}
        mod.summarize(summary, lane)
        out.append({k: (None if isinstance(v, float) and math.isnan(v) else v) for k, v in sorted(summary.items())})
    return out


if __name__ == "__main__":
    import pathlib
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.test_summary_consumer import paired_config1_bamqc
    mod = load_reference_module()
    with tempfile.TemporaryDirectory() as td:
        path, n = paired_config1_bamqc(pathlib.Path(td)) # config 1 reduced to complete pairs, written by the oracle
        res = {"source": "DecodeGenetics/BamQC bamqc_summary.py read_bamqc_output + summarize on the paired config-1 output of the oracle",
               "pairs": n, "lanes": summarize_file(mod, path)}
    json.dump(res, open(os.path.join(HERE, "config1_summary.json"), "w"), indent=1, sort_keys=True)
    print(len(res["lanes"]), "lane(s);", len(res["lanes"][0]), "summary fields")
