"""CPU: the in-tree C-ABI library loads and exports every symbol include/bamqc.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

from bamqc_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            h = open(os.path.join(inc, f)).read()
            h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
            names |= set(re.findall(r"\b(bqc_[a-z_0-9]+)\s*\(", h))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libbamqc_gpu.so does not export %s" % n
    missing = set(names) - set(_lib._SIGNATURES)
    assert not missing, "python binding lacks %s" % sorted(missing)


def test_abi_version_and_struct_sizes():
    lib = _lib.load()
    assert lib.bqc_abi_version() == 2
    assert ctypes.sizeof(_abi.Options) % 8 == 0
    assert ctypes.sizeof(_abi.LaneCounts) > 0


def test_create_without_gpu_fails_loudly_or_succeeds_on_gpu():
    # No CPU fallback: on a box without a HIP device creation must fail with BQC_ERR_DEVICE.
    import torch
    from bamqc_amd import Aggregator, BamQCError
    if torch.cuda.is_available():
        Aggregator(n_refs=1).close()
        return
    try:
        Aggregator(n_refs=1)
    except BamQCError as e:
        assert e.code == 2
    else:
        raise AssertionError("bqc_create succeeded without a GPU")


def test_product_does_not_reference_oracle():
    # the product tree must never import / link / execute anything under oracle/
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "bamqc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"oracle/|liboracle|orc_[a-z]+|import\s+.*oracle|from\s+.*oracle", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_program_args_are_checked_without_running_anything():
    """bqc_program_args (what `bamqualcheck --gpus N` asks before it forks): the program's own parser, no GPU call — the input path
    wherever it stands on the line, usage errors as status 1, help as 2."""
    lib = _lib.load()

    def ask(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"bamqualcheck", *[a.encode() for a in args])
        buf = ctypes.create_string_buffer(4096)
        rc = lib.bqc_program_args(len(args) + 1, argv, buf, 4096)
        return rc, buf.value.decode()

    assert ask("-r", "g.fa", "-o", "x.bamqc", "in.bam") == (0, "in.bam")
    assert ask("-r", "g.fa", "in.bam", "-o", "looks_like_input.bam") == (0, "in.bam")
    assert ask("-r", "g.fa", "-o", "x.bamqc", "-") == (0, "-")
    assert ask("-r", "g.fa", "-o", "x.bamqc", "--no-such-option", "in.bam")[0] == 1
    assert ask("-r", "g.fa", "-o", "x.bamqc")[0] == 1              # no input
    assert ask("-r", "g.fa", "-o", "x.bamqc", "in.txt")[0] == 1    # not *.bam / *.sam / -
    assert ask("--help")[0] == 2
