"""Sanitizer leg (the reference fuzzes its decoder; here AddressSanitizer + UBSan run over the C / C++ restatements
that parse untrusted bytes): the product's host front-end, its device-entropy planner and compact-transport writer
(csrc/host/frontend.cpp, csrc/compact.hpp, csrc/image_job.cpp) and the oracle's front-end, on the reference's reftest,
bench and crash corpora plus truncated and bit-flipped variants.  Pass = no sanitizer report (decode errors are outcomes)."""
import glob
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CSRC = os.path.join(ROOT, "jpeg-decoder_amd", "csrc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("asan") / "asan_driver")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    objs = []
    for src in ("oracle_pixels.c", "oracle_front.c", "oracle_batch.c"):
        o = out + "_" + src + ".o"
        subprocess.check_call(["gcc", "-std=c11", "-fwrapv", "-fno-strict-aliasing", *san, "-c", os.path.join(ROOT, "oracle", src), "-o", o])
        objs.append(o)
    subprocess.check_call(["g++", "-std=c++17", *san, "-DJPGPU_HOST_EMULATION", "-I", os.path.join(ROOT, "tests", "emu"), "-include", "hip_shim.hpp",
                           os.path.join(ROOT, "tests", "emu", "asan_driver.cpp"), os.path.join(CSRC, "host", "frontend.cpp"),
                           os.path.join(CSRC, "image_job.cpp"), *objs, "-o", out, "-lpthread", "-lm"])
    return out


def _corpus():
    files = []
    for pat in ("reftest/*.jp*g", "reftest/mozilla/*.jp*g", "benches/*.jpg", "crashtest/**/*", "icc/*"):
        files += [f for f in glob.glob(os.path.join(GOLDEN, pat), recursive=True) if os.path.isfile(f)]
    return sorted(set(files))


def _run(driver, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver] + files, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


@pytest.mark.timeout(1200)
def test_corpora_under_asan_and_ubsan(driver):
    files = _corpus()
    assert len(files) > 40
    out = _run(driver, files)
    assert "decoded" in out


@pytest.mark.timeout(1200)
def test_mutated_streams_under_asan_and_ubsan(driver, tmp_path):
    """Truncations and random byte flips of small corpus files: the parsers must refuse or decode, never read or write
    out of bounds."""
    rng = random.Random(2024)
    small = [f for f in _corpus() if 16 <= os.path.getsize(f) < 60000][:24]
    files = []
    for i, f in enumerate(small):
        data = bytearray(open(f, "rb").read())
        for j in range(6):
            d = bytearray(data)
            if j < 2:
                d = d[: max(4, rng.randrange(len(d)))]
            else:
                for _ in range(rng.randrange(1, 12)):
                    d[rng.randrange(2, len(d))] = rng.randrange(256)
            p = tmp_path / f"m{i}_{j}.jpg"
            p.write_bytes(bytes(d))
            files.append(str(p))
    _run(driver, files)
