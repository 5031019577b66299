"""The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer.

tests/planner/build.py compiles the host half of every source of libgscan_hip.so with -fsanitize=address,undefined (no
device code, device calls are no-ops, launches are checked against the hardware limits and counted) and
tests/planner/driver.hip calls the C ABI over a list of shapes: the benchmark configurations, the degenerate / limit
shapes of tools/fuzz_parity.py --extremes, and random shapes over the whole range the reference's flags accept
(--wide: hidden sizes 1..1024, grids to 12 x 12, commands to 128 tokens, 1-3 encoder layers).  Runs on the CPU; nothing
here touches a GPU."""
import os
import random
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "planner"))

# (H, He, E, k, Co, cond, aux, bi, layers, B, G, L, T, Vi, V, C)
BENCH = [
    (100, 100, 25, 7, 50, 1, 0, 1, 1, 256, 6, 10, 20, 21, 9, 16),      # S1 = BASELINE.json configs[1]
    (100, 100, 25, 13, 50, 1, 0, 1, 1, 256, 6, 10, 120, 17, 8, 16),    # S3 = configs[3]
    (100, 100, 25, 7, 50, 1, 1, 1, 1, 256, 6, 10, 20, 21, 9, 16),      # S4 = configs[4] per-GPU shard
    (20, 20, 5, 7, 50, 1, 0, 1, 1, 4, 4, 7, 10, 14, 6, 15),            # S0 = the README's demo
    (100, 100, 25, 7, 50, 1, 0, 1, 1, 2048, 6, 10, 20, 21, 9, 16),     # a whole configs[2] batch on one device
]
EXTREMES = [(100, 100, 25, 7, 50, 1, 0, 1, 1, 1, 2, 1, 1), (100, 100, 25, 7, 50, 1, 1, 1, 1, 2, 2, 2, 1),
            (100, 100, 25, 7, 50, 1, 0, 1, 1, 1, 6, 64, 2), (100, 100, 25, 7, 50, 1, 0, 1, 1, 2, 8, 10, 3),
            (100, 100, 25, 7, 50, 1, 0, 1, 1, 1, 8, 64, 2), (256, 256, 64, 7, 50, 1, 1, 1, 2, 2, 12, 128, 3),
            (255, 255, 63, 7, 50, 1, 0, 1, 1, 2, 3, 5, 3), (1, 1, 1, 1, 1, 1, 1, 0, 1, 2, 2, 2, 2),
            (100, 100, 25, 13, 50, 0, 0, 0, 1, 3, 2, 1, 40), (4, 4, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2),
            (100, 128, 64, 7, 50, 1, 0, 1, 2, 2, 6, 10, 5), (64, 64, 33, 3, 70, 1, 1, 1, 3, 5, 7, 9, 3),
            (100, 100, 25, 7, 200, 1, 0, 1, 1, 2, 6, 10, 20), (96, 100, 25, 7, 50, 1, 0, 1, 1, 257, 6, 10, 4)]
# beyond every limit: each must come back as a message, not as a crash
REJECTED = [(1025, 100, 25, 7, 50, 1, 0, 1, 1, 2, 6, 10, 4, 21, 9, 16), (100, 4096, 25, 7, 50, 1, 0, 1, 1, 2, 6, 10, 4, 21, 9, 16),
            (100, 100, 25, 6, 50, 1, 0, 1, 1, 2, 6, 10, 4, 21, 9, 16), (100, 100, 25, 7, 50, 1, 0, 1, 5, 2, 6, 10, 4, 21, 9, 16),
            (100, 100, 25, 7, 50, 1, 0, 1, 1, 0, 6, 10, 4, 21, 9, 16), (100, 100, 25, 7, 50, 1, 0, 1, 1, 60000, 6, 10, 100, 21, 9, 16),
            (100, 100, 1300, 7, 50, 1, 0, 1, 1, 2, 6, 10, 4, 21, 9, 16), (100, 100, 25, 7, 50, 1, 0, 1, 1, 2, 70, 10, 4, 21, 9, 16)]


def _wide(cases: int, seed: int):
    rng = random.Random(seed)
    out = []
    for _ in range(cases):
        H = rng.choice([rng.randint(1, 256), rng.choice([128, 200, 256, 600, 1024]), rng.choice(list(range(4, 101, 4)))])
        He = rng.choice([rng.randint(1, 256), rng.choice([128, 200, 256]), rng.choice(list(range(4, 129, 4)))])
        out.append((H, He, rng.choice([4, 5, 8, 25, 64]), rng.choice([1, 3, 5, 7, 13]), rng.choice([8, 20, 50, 70, 200]),
                    int(rng.random() < 0.6), int(rng.random() < 0.5), int(rng.random() < 0.7), rng.choice([1, 1, 2, 3]),
                    rng.choice([1, 2, 3, 5, 9, 64, 256, 300]), rng.choice([2, 3, 4, 6, 8, 9, 10, 12]),
                    rng.choice([1, 2, 7, 10, 17, 40, 65, 100, 128]), rng.choice([1, 2, 3, 10, 17, 33, 120]),
                    rng.choice([8, 14, 21]), rng.choice([2, 5, 6, 9, 17, 64]), rng.choice([15, 16])))
    return out


@pytest.fixture(scope="module")
def planner():
    import shutil
    # the host-only build needs hipcc (for the HIP headers and --offload-host-only), gcc's sanitizer runtimes and nm: a CPU
    # box without ROCm skips instead of erroring
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not found: the sanitizer build of the host planners needs the ROCm toolchain")
    for tool in ("gcc", "nm"):
        if not shutil.which(tool):
            pytest.skip(f"{tool} not found: the sanitizer build of the host planners needs it")
    import build as planner_build
    return planner_build.build()


@pytest.mark.parametrize("env", [{}, {"GSCAN_DETERMINISTIC": "1"}, {"GSCAN_DECODER_ANY": "1", "GSCAN_ENCODER_ANY": "1"},
                                 {"GSCAN_GEMM_MT": "1", "GSCAN_FORWARD_STREAMS": "3", "GSCAN_FUSED_PROLOGUE": "0"}],
                         ids=["default", "deterministic", "streaming_kernels", "macro_tiles_three_streams"])
def test_host_planners_are_clean_under_asan_and_ubsan(planner, tmp_path, env):
    shapes = BENCH + [s + (14, 9, 16) for s in EXTREMES] + REJECTED + _wide(120, seed=5)
    path = tmp_path / "shapes.txt"
    path.write_text("\n".join(" ".join(str(v) for v in s) for s in shapes) + "\n")
    r = subprocess.run([planner, str(path)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
                                **env))
    tail = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0, tail                                   # a sanitizer report or a launch beyond the hardware limits
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    summary = r.stdout.strip().splitlines()[-1]
    assert summary.startswith(f"shapes {len(shapes)},"), summary
    launches = int(summary.split("kernel launches planned ")[1].split(",")[0])
    assert launches > 20 * len(BENCH), summary                        # the steps really were sequenced
    # the shapes beyond the limits are refused with a message each (workspace_bytes says why)
    assert r.stdout.count("rejected workspace_bytes") >= len(REJECTED), r.stdout[-3000:]
