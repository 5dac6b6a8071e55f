"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/medtok_vq.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "medtok_vq.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(medtok_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from medtok_amd import _lib
    from medtok_amd.csrc import build as builder
    builder.build()
    lib = ctypes.CDLL(str(_lib.library_path()))
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in medtok_vq.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"


def test_library_loads_and_reports_version():
    from medtok_amd import _lib
    lib = _lib.load()
    assert lib.medtok_abi_version() == _lib.ABI_VERSION
    # size queries are host-only arithmetic: safe without a GPU
    assert lib.medtok_search_workspace_bytes(1000, 8192, 768, 5, 0) > 0
    assert lib.medtok_search_workspace_bytes(3000000, 16384, 768, 5, _lib.PATH_F32_MFMA) == 256   # exact path, single split: no scratch
    assert lib.medtok_search_workspace_bytes(100000, 8192, 768, 5, _lib.PATH_F16_FILTER) > 100000 * 768 * 2   # fp16 copy + candidate lists
    assert lib.medtok_search_workspace_bytes(100, 64, 32, 5, _lib.PATH_AUTO) == lib.medtok_search_workspace_bytes(100, 64, 32, 5, _lib.PATH_F32_MFMA)
    assert lib.medtok_ema_stats_workspace_bytes(100000, 8192) > 8 * 100000
    assert lib.medtok_usage_workspace_bytes(300000, 21000) >= 300000 * 4


def test_batched_search_planning_is_host_only():
    """medtok_soft_vq_multi_eligible / ..._workspace_bytes are host arithmetic: which searches a batched call takes (the exact path, at
    most 4096 rows) and what scratch it needs -- checked without a GPU"""
    from medtok_amd import _lib
    lib = _lib.load()
    assert lib.medtok_soft_vq_multi_eligible(256, 7000, 64, 5) == 1 and lib.medtok_soft_vq_multi_eligible(512, 21000, 64, 5) == 1
    assert lib.medtok_soft_vq_multi_eligible(4097, 21000, 64, 5) == 0            # too many rows
    assert lib.medtok_soft_vq_multi_eligible(4096, 49152, 768, 5) == 0           # the fp16-filter path's territory
    assert lib.medtok_soft_vq_multi_eligible(256, 7000, 62, 5) == 0              # d % 4
    descs = (_lib.SearchDesc * 3)()
    for i, (n, k) in enumerate(((512, 21000), (256, 7000), (256, 7000))):
        descs[i] = _lib.SearchDesc(0, n, 0, 0, k, 0, 0, 0, 0, 0, 0, 0)
    need = lib.medtok_soft_vq_forward_multi_workspace_bytes(descs, 3, 64, 5)
    assert need > 512 * 4 + 2 * 256 * 4                                          # at least the squared norms
    assert lib.medtok_soft_vq_forward_multi_workspace_bytes(descs, 7, 64, 5) == 0   # more searches than a call takes
    assert lib.medtok_usage_multi_workspace_bytes(300000, 21000, 3) >= 300000 * 4 + 3 * 21001


def test_code_object_targets_gfx950_only():
    from medtok_amd import _lib
    blob = _lib.library_path().read_bytes()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_product_does_not_import_the_oracle():
    for p in (ROOT / "medtok_amd").rglob("*.py"):
        src = p.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f"{p} mentions the oracle"


def test_header_macros_agree_with_the_python_binding():
    """The bit layout of the `path` / `variant` arguments is written twice (include/medtok_vq.h and medtok_amd/_lib.py, ops.py):
    the two must say the same thing.  Host-only: the workspace query decodes the plan bits (a search with more code splits needs
    more candidate lists, i.e. more workspace; rows of <= 32 elements are padded to 64 columns on either filter kernel)."""
    from medtok_amd import _lib, ops
    text = (ROOT / "include" / "medtok_vq.h").read_text()
    macro = lambda name: re.search(r"#define\s+" + name + r"(?:\([^)]*\))?\s+(.+?)\s*(?:/\*|$)", text, re.M).group(1)
    assert int(macro("MEDTOK_PATH_MASK"), 0) == 0xF
    assert int(macro("MEDTOK_ATTENTION_F32_KEYS"), 0) == ops.ATTENTION_F32_KEYS
    for m in re.finditer(r"#define\s+MEDTOK_PATH_(AUTO|F32_MFMA|F16_FILTER)\s+(\d+)", text):
        assert getattr(_lib, "PATH_" + m.group(1)) == int(m.group(2))
    ev = lambda name, arg: eval(macro(name).replace("(s)", f"({arg})").replace("(on)", f"({arg})").replace("?", " and ").replace(":", " or "))  # noqa: S307
    assert _lib.plan_path(_lib.PATH_F16_FILTER, filter_splits=5) == _lib.PATH_F16_FILTER | ev("MEDTOK_PLAN_FILTER_SPLITS", 5)
    assert _lib.plan_path(0, filter_xcd=True) == ev("MEDTOK_PLAN_FILTER_XCD", 1) and _lib.plan_path(0, filter_xcd=False) == ev("MEDTOK_PLAN_FILTER_XCD", 0)
    assert _lib.plan_path(0, filter_tail=True) == ev("MEDTOK_PLAN_FILTER_TAIL", 1) and _lib.plan_path(0, filter_tail=False) == ev("MEDTOK_PLAN_FILTER_TAIL", 0)
    assert _lib.plan_path(0, search_max_splits=7) == ev("MEDTOK_PLAN_SEARCH_MAX_SPLITS", 7)
    assert _lib.plan_path(0, filter_rows64=True) == ev("MEDTOK_PLAN_FILTER_ROWS64", 1) and _lib.plan_path(0, filter_rows64=False) == ev("MEDTOK_PLAN_FILTER_ROWS64", 0)
    lib = _lib.load()
    ws = lambda n, k, d, path: lib.medtok_search_workspace_bytes(n, k, d, 5, path)
    f = _lib.PATH_F16_FILTER
    assert ws(100000, 8192, 768, _lib.plan_path(f, filter_splits=4)) > ws(100000, 8192, 768, _lib.plan_path(f, filter_splits=1))
    assert _lib.plan_path(0, filter_rows64="wide") == int(macro("MEDTOK_PLAN_FILTER_ROWS64_WIDE").replace("(", "").replace(")", "").split("<<")[0]) << 28
    # D <= 64: the 128 x 64 wave-tile kernel keeps four candidate lists per row and split like the general one, the 128 x 32 one two
    assert ws(100000, 8192, 64, _lib.plan_path(f, filter_rows64="wide", filter_splits=2)) == ws(100000, 8192, 64, _lib.plan_path(f, filter_rows64=False, filter_splits=2))
    assert ws(100000, 8192, 64, _lib.plan_path(f, filter_rows64=True, filter_splits=2)) < ws(100000, 8192, 64, _lib.plan_path(f, filter_rows64=False, filter_splits=2))
    assert ws(100000, 8192, 16, _lib.plan_path(f, filter_splits=2)) == ws(100000, 8192, 64, _lib.plan_path(f, filter_splits=2))      # D <= 32: 64 columns
