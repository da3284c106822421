"""ctypes access to libbthost.so, the C++ host layer (classes mirroring the reference's interface) above libbtgpu.so."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libbthost.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: run bayestyper_amd/host/build.sh (or __graft_entry__.build())")
# libbthost links libbtgpu.so; load that first so the dependency resolves from the package directory
C.CDLL(os.path.join(os.path.dirname(_HERE), "libbtgpu.so"), mode=C.RTLD_GLOBAL)
dll = C.CDLL(LIB_PATH)
