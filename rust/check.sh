#!/bin/bash
# Apply the binding to a copy of image-rs/jpeg-decoder v0.3.2 and, where a Rust toolchain exists, compile it.
#   rust/check.sh /path/to/jpeg-decoder [dir holding libjpgpu.so]
# Exit status: 0 = patches applied and `cargo check --features hip` passed; 3 = patches applied, no cargo on this machine
# (NOTHING was compiled: the state of this repository's own build image); anything else = failure.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=${1:?path to a checkout of image-rs/jpeg-decoder v0.3.2}
LIBDIR=${2:-$HERE/../jpeg-decoder_amd}
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$SRC" "$WORK/crate"
cd "$WORK/crate"
for p in worker_mod decoder cargo_toml; do patch -p1 --quiet < "$HERE/$p.patch"; done
cp "$HERE/src/worker/hip.rs" src/worker/hip.rs
cp "$HERE/build.rs" build.rs
echo "patches applied to a copy of $SRC"
if command -v cargo > /dev/null 2>&1; then
    JPGPU_LIB_DIR="$LIBDIR" cargo check --features hip,platform_independent
    echo "cargo check --features hip passed"
else
    echo "no cargo on this machine: nothing was compiled"
    exit 3
fi
