"""Test-only host build of the lane-serial kernel code (see sim_*.cpp)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhost_sim.so")
SRCS = ["sim_inflate.cpp", "sim_deflate.cpp", "sim_chain.cpp"]


def lib():
    if os.environ.get("ZD_HOST_SIM_LIB"):  # another build of the same sources (tests/test_sanitizers.py: address + undefined sanitizers)
        return _bind(C.CDLL(os.environ["ZD_HOST_SIM_LIB"]))
    deps = [os.path.join(HERE, s) for s in SRCS]
    csrc = os.path.join(HERE, "..", "..", "zipc_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    deps += [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall",
                               "-Wno-unknown-pragmas", "-I", HERE, "-o", LIB] + [os.path.join(HERE, s) for s in SRCS])
    return _bind(C.CDLL(LIB))


def _bind(L):
    L.sim_inflate.restype = C.c_int
    L.sim_inflate.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_int,
                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int]
    L.sim_deflate.restype = C.c_int
    L.sim_deflate.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                              C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    L.sim_huff_lengths.restype = C.c_int
    L.sim_huff_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sim_crc_advance.restype = C.c_uint32
    L.sim_crc_advance.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    L.sim_chain.restype = None
    L.sim_chain.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
    L.sim_chain_serial.restype = None
    L.sim_chain_serial.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p]
    return L
