"""Many decodes on many threads: the reference pins that with tests/rayon-2.rs:14-20 (1024 threads, each decoding the same
image on the global rayon pool).  Here: 64 Python threads x Decoder(data).decode() (ctypes releases the GIL inside the
library: the worker / pipeline pools behind their mutexes in csrc/host/decoder_api.cpp are really entered concurrently),
next to 4 Pipelines decoding batches on the same device, every result byte-exact against the oracle; and the reference's test
at its own size (1,024 threads) against the library's cap on concurrently held device contexts."""
import os
import threading

import numpy as np
import pytest

import oracle as O
import refimages as R

pytestmark = pytest.mark.gpu


def _images():
    names = [os.path.join(R.REFTEST, "mozilla", "jpg-size-33x33.jpg"), os.path.join(R.REFTEST, "mjpeg.jpg"),
             os.path.join(R.REFTEST, "progressive3.jpg"), os.path.join(R.REFTEST, "rgb.jpg"),
             os.path.join(R.GOLDEN, "benches", "tower.jpg"), os.path.join(R.GOLDEN, "benches", "tower_grayscale.jpg"),
             os.path.join(R.GOLDEN, "benches", "tower_progressive.jpg"), os.path.join(R.REFTEST, "mozilla", "jpg-cmyk-2.jpg")]
    out = []
    for n in names:
        data = open(n, "rb").read()
        out.append((os.path.basename(n), data, O.decode(data).pixels))
    return out


@pytest.mark.timeout(600)
def test_64_decoder_threads_and_4_pipelines_on_one_device():
    import jpeg_decoder_amd as J
    assert J.device_count() >= 1
    imgs = _images()
    # a big sequential image as well: it takes the one-image pipeline route inside Decoder.decode (pooled pipelines)
    big = open(os.path.join(R.GOLDEN, "benches", "large_image.jpg"), "rb").read()
    imgs.append(("large_image.jpg", big, O.decode(big).pixels))
    errors, lock = [], threading.Lock()
    start = threading.Barrier(64 + 4)

    def decoder_thread(t):
        try:
            start.wait()
            for rep in range(6):
                name, data, want = imgs[(t + rep) % len(imgs)]
                got = J.Decoder(data).decode()
                if not np.array_equal(got, want):
                    raise AssertionError(f"thread {t} rep {rep}: {name} differs from the oracle")
        except Exception as e:  # noqa: BLE001 - collected and re-raised in the main thread
            with lock:
                errors.append(repr(e))

    def pipeline_thread(t):
        try:
            p = J.Pipeline(threads=4)
            start.wait()
            for rep in range(3):
                batch = [imgs[(t + rep + k) % len(imgs)] for k in range(12)]
                outs = p.decode([b[1] for b in batch], device_entropy=(t % 2 == 0))
                for (name, _d, want), got in zip(batch, outs):
                    if isinstance(got, Exception) or not np.array_equal(got, want):
                        raise AssertionError(f"pipeline {t} rep {rep}: {name}: {got!r}")
            p.close()
        except Exception as e:  # noqa: BLE001
            with lock:
                errors.append(repr(e))

    threads = [threading.Thread(target=decoder_thread, args=(t,)) for t in range(64)] + \
              [threading.Thread(target=pipeline_thread, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(500)
        assert not th.is_alive(), "a decoding thread hung"
    assert not errors, errors[:5]


@pytest.mark.timeout(900)
def test_1024_decoding_threads_like_the_reference_rayon_2_test():
    """tests/rayon-2.rs:14-20 at its own size: 1,024 threads, each `Decoder::new(progressive3.jpg).decode()`.  The reference's
    threads queue up behind a two-thread rayon pool; here they queue up behind the library's cap on concurrently held device
    contexts (64 by default, JPGPU_MAX_CONCURRENT_DECODES) — a thousand decodes do not mean a thousand streams and buffer
    sets on the device.  Every result is checked (the reference only unwraps).  JPGPU_TEST_RAYON2_THREADS scales the test."""
    import jpeg_decoder_amd as J
    n_threads = int(os.environ.get("JPGPU_TEST_RAYON2_THREADS", "1024"))
    data = open(os.path.join(R.REFTEST, "progressive3.jpg"), "rb").read()
    want = O.decode(data).pixels
    errors, lock = [], threading.Lock()
    done = [0]

    def body(t):
        try:
            got = J.Decoder(data).decode()
            if not np.array_equal(got, want):
                raise AssertionError(f"thread {t}: differs from the oracle")
            with lock:
                done[0] += 1
        except Exception as e:  # noqa: BLE001
            with lock:
                errors.append(repr(e))

    old = threading.stack_size(256 * 1024)  # (a thousand default-sized stacks are 8 GB of address space)
    try:
        threads = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(800)
            assert not th.is_alive(), "a decoding thread hung"
    finally:
        threading.stack_size(old)
    assert not errors, errors[:5]
    assert done[0] == n_threads


@pytest.mark.timeout(600)
def test_two_pipelines_walk_progressive_frames_at_the_same_time(monkeypatch):
    """Progressive frames on the device are walked a WAVE per scan, and a wave that waits for the scan it depends on holds its slot
    (csrc/huff_prog_wave.hpp).  Two pipelines walking at the same moment — 2 x 2,560 frames x 10 scans = 51 k waves, six times what
    the device holds at once — must both come through: the launch order keeps producers in front of their consumers on every XCD, so
    nothing can wait for a wave that is not resident or done; every frame must be right, whether its waves ran as planned or gave up
    waiting and left the frame to the host."""
    import jpeg_decoder_amd as J
    monkeypatch.setenv("JPGPU_PIPE_PROG_DEVICE_PERCENT", "100")
    tower = open(os.path.join(R.GOLDEN, "benches", "tower_progressive.jpg"), "rb").read()
    want = O.decode(tower).pixels
    n = 2560
    problems, on_device = [], []

    def work(t):
        try:
            p = J.Pipeline(threads=8)
            for _ in range(3):
                res = p.decode([tower] * n, device_entropy=True, download=False)
                if any(isinstance(x, Exception) for x in res):
                    problems.append((t, "error"))
                on_device.append(int(p.timings()["images_device_progressive"]))
                for i in (0, 1, 63, 64, n // 2, n - 2, n - 1):
                    if p.download(i).tobytes() != np.ascontiguousarray(want).tobytes():
                        problems.append((t, i))
            p.close()
        except Exception as e:  # noqa: BLE001
            problems.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=240)
    assert not any(th.is_alive() for th in threads), "a pipeline hangs"
    assert not problems, problems[:4]
    assert on_device and min(on_device) == n, on_device
