"""The crate-side binding under rust/ (ADVICE r2): its patches must apply to the reference crate.  Compiling them needs a Rust
toolchain, which this image does not have — rust/check.sh does both where cargo exists and says so where it does not.  Runs only
where the reference checkout is present (the build container), never on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "worker")) or not shutil.which("patch"),
                    reason="needs the reference checkout and patch(1)")
def test_patches_apply_to_the_reference_crate():
    r = subprocess.run(["bash", os.path.join(ROOT, "rust", "check.sh"), REFERENCE], capture_output=True, text=True)
    assert "patches applied" in r.stdout, r.stdout + r.stderr
    # 0: compiled as well; 3: no cargo here, nothing compiled (the documented state of this image)
    assert r.returncode in (0, 3), r.stdout + r.stderr
    if r.returncode == 3:
        assert "nothing was compiled" in r.stdout
