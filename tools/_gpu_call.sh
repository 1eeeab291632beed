cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu.sh r5k tests
O=gpurun_out/r5k
python tools/prog_calls.py --images 256 | tail -2
python tools/prog_calls.py --images 4096 | tail -2
python tools/prog_calls.py --images 4096 --distinct | tail -2
JPGPU_PIPE_HOST_LIGHT=1 bash tools/gpu.sh r5k fuzz:150
