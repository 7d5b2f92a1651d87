"""Sanitizers where they can run: CPU builds only (GPU AddressSanitizer and XNACK runs are not available on this pool).

* the many-stream forms' host pipeline -- the staging pools, the thread that feeds the device, the thread that takes
  results back (zipc_amd/csrc/host_pipeline.h, the very header api.hip compiles) -- under ThreadSanitizer and under
  Address + UndefinedBehaviour sanitizers, with host threads standing in for the device (tests/host_sim/pipeline_sim.cpp);
* the oracle (oracle/zd_oracle.c), the host models of the kernels (tests/host_sim/sim_*.cpp: inflate_lane.h,
  inflate_span.h, deflate_lane.h compiled for the host) and the C++ host layer (zipc_amd/host/*.cpp: ZIP container, member
  glue) rebuilt with -fsanitize=address,undefined and driven through bounded parts of their own test modules, in a
  child process with the sanitizer's runtime preloaded.
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SIM = os.path.join(HERE, "host_sim")
BUILD = os.path.join(SIM, "build_san")


def _san_runtime(name):
    p = subprocess.run(["g++", "-print-file-name=lib%s.so" % name], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("lib%s.so not found beside g++" % name)
    return p


def _build(out, cmd, deps):
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(cmd)
    return out


def _pipeline(kind, flags):
    src = os.path.join(SIM, "pipeline_sim.cpp")
    hdr = os.path.join(ROOT, "zipc_amd", "csrc", "host_pipeline.h")
    out = os.path.join(BUILD, "pipeline_" + kind)
    return _build(out, ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-DZD_HOST_PLAIN_COPY"] + flags + ["-o", out, src, "-lpthread"],
                  [src, hdr, os.path.join(ROOT, "include", "zipc_hip.h")])


def test_host_pipeline_under_thread_sanitizer():
    _san_runtime("tsan")
    exe = _pipeline("tsan", ["-fsanitize=thread"])
    # (no fork leg: ThreadSanitizer cannot follow threads made after a fork of a threaded process)
    r = subprocess.run([exe, "11", "12", "nofork"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
    assert r.returncode == 0 and "pipeline_sim: 12 calls ok" in r.stdout and "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]


def test_host_pipeline_under_address_sanitizer_with_fork_and_failures():
    _san_runtime("asan")
    exe = _pipeline("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = subprocess.run([exe, "12", "40"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "pipeline_sim: 40 calls ok" in r.stdout and "Sanitizer" not in r.stdout, r.stdout[-4000:]


def test_host_pipeline_with_the_streaming_copies():
    """the product's own copies (stores around the cache), plain build: the sanitizers' builds use memcpy in their place"""
    exe = _pipeline("plain", ["-O2"])
    r = subprocess.run([exe, "13", "60"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "pipeline_sim: 60 calls ok" in r.stdout, r.stdout[-4000:]


def _pytest_under_asan(extra_env, args, timeout=900):
    # (libstdc++ beside it: the interpreter is not linked against it, and the sanitizer's __cxa_throw interceptor looks its real
    # symbol up when the runtime starts)
    asan = _san_runtime("asan") + " " + subprocess.run(["g++", "-print-file-name=libstdc++.so.6"], stdout=subprocess.PIPE, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-6000:]
    return r.stdout


def test_oracle_under_address_sanitizer():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    out = _pytest_under_asan({"ZD_ORACLE_LIB": os.path.join(ROOT, "oracle", "libzd_oracle_asan.so")}, ["tests/test_oracle_pins.py"])
    assert " passed" in out


def test_kernel_host_models_under_address_sanitizer():
    """inflate_lane.h / inflate_span.h / deflate_lane.h as the host compiles them, on the fast cases of their own module"""
    srcs = [os.path.join(SIM, s) for s in ("sim_inflate.cpp", "sim_deflate.cpp", "sim_chain.cpp")]
    csrc = os.path.join(ROOT, "zipc_amd", "csrc")
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + [os.path.join(SIM, f) for f in os.listdir(SIM) if f.endswith(".h")]
    out = os.path.join(BUILD, "libhost_sim_asan.so")
    _build(out, ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-fsanitize=address,undefined",
                 "-fno-sanitize-recover=undefined", "-I", SIM, "-o", out] + srcs, deps)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    fast = "golden_streams or accept_reject or lane_logic_fuzz or header_fuzz or wide_turn or header_search or crc_combination or two_queues or chain_round or segments_model_bytes_equal_oracle-64"
    out = _pytest_under_asan({"ZD_HOST_SIM_LIB": out, "ZD_ORACLE_LIB": os.path.join(ROOT, "oracle", "libzd_oracle_asan.so")},
                             ["tests/test_host_sim.py", "-k", fast])
    assert " passed" in out


def test_host_layer_under_address_sanitizer():
    """zipc_amd/host/*.cpp (ZIP container, member glue, the C view) on the codec-free container tests"""
    if not os.path.exists(os.path.join(ROOT, "zipc_amd", "lib", "libzipc_hip.so")):
        pytest.skip("libzipc_hip.so is not built")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zipc_amd", "host"), "asan"])
    out = _pytest_under_asan({"ZIPC_HOST_LIB": os.path.join(ROOT, "zipc_amd", "lib", "libzipc_host_asan.so")},
                             ["tests/test_zipc_container.py", "-m", "not gpu"])
    assert " passed" in out
