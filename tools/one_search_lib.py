import ctypes, sys
import torch  # load torch (and its HIP runtime) before any other HIP client library
sys.path.insert(0, ".")
from medtok_amd import _lib
lib = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_LOCAL)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
_lib._lib = lib
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open("tools/one_search.py").read())
