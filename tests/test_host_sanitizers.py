"""The library's HOST code under ThreadSanitizer and under AddressSanitizer + UndefinedBehaviourSanitizer.

runtime.cpp (pool, waits, profiling scopes), capi.cpp (the stream scheduler of mm3d_estimate_maps_transforms with its worker
threads and rand() state table, the shard driver, params / enums / error paths), host_pipeline.cpp (RANSAC / SAC-IA replays,
pair driver, pose graph) and linalg.cpp are compiled UNCHANGED by clang's host pass with the sanitizer on and linked against a
fake HIP runtime and a fake device layer (tests/host_san/: "device" memory is host memory, every stage returns a cheap
deterministic placeholder).  The driver (tests/host_san/san_main.cpp) runs whole jobs on 1, 3, 5, 8 and 16 streams and on a
world of three emulated ranks and requires identical bits, plus the degenerate inputs of the reference's gtests
(R/test/test_map_merging.cpp:9-40).  No GPU is needed; the product itself still has no CPU path."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = os.environ.get("CLANG", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.mark.parametrize("kind", ["thread", "address"])
def test_host_code_under_sanitizer(kind):
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("no clang with HIP support here")
    subprocess.check_call([os.path.join(HERE, "host_san", "build.sh"), kind], stdout=subprocess.DEVNULL)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("MM3D_FEATURE_WORKERS", None)
    r = subprocess.run([os.path.join(HERE, "host_san", "_build", "san_" + kind)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "host sanitizer driver ok" in r.stdout
    for word in ("ThreadSanitizer", "AddressSanitizer", "LeakSanitizer", "runtime error"):
        assert word not in r.stderr, r.stderr[-6000:]
