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
