"""Shared helper: push the same batches through the HIP library (C ABI) and the oracle."""
import numpy as np

from bamqc_amd import _abi
from tests import synth
from tests.oracle_lib import Oracle


def run_oracle(cols_list, refs, **opts):
    o = Oracle(**opts)
    for i, r in enumerate(refs or []):
        if r is not None:
            o.reference(i, r)
    for cols in cols_list:
        rc = o.process(cols)
        if rc:
            return rc, None, o
    return 0, o.finalize(), o


def run_gpu(cols_list, refs, **opts):
    from bamqc_amd import Aggregator, BamQCError
    a = Aggregator(**opts)
    for i, r in enumerate(refs or []):
        if r is not None:
            a.set_reference(i, r)
    try:
        for cols in cols_list:
            a.submit(cols)
        return 0, a.finalize(), a
    except BamQCError as e:
        return e.code, None, a


def assert_parity(cols_list, refs, **opts):
    if isinstance(cols_list, dict):
        cols_list = [cols_list]
    n_refs = max(1, len(refs) if refs else 1)
    opts.setdefault("n_refs", n_refs)
    rc_o, co, o = run_oracle(cols_list, refs, **opts)
    rc_g, cg, a = run_gpu(cols_list, refs, **opts)
    assert rc_o == rc_g, "error code: oracle %d, gpu %d" % (rc_o, rc_g)
    if rc_o == 0:
        d = _abi.diff_counts(co, cg)
        assert not d, "\n".join(d[:20])
    return co, cg, o, a


def split(cols, cuts):
    edges = [0] + list(cuts) + [len(cols["flag"])]
    return [synth.slice_batch(cols, edges[i], edges[i + 1]) for i in range(len(edges) - 1) if edges[i + 1] > edges[i]]
