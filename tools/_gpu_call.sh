cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
for v in engine kernel engine kernel; do
  if [ $v = kernel ]; then export JPGPU_UPLOAD_BY_KERNEL=1; else unset JPGPU_UPLOAD_BY_KERNEL; fi
  for n in 4096 256; do echo "== $v $n"; python tools/pipe_calls.py --images $n --calls 8 2>&1 | tail -1; done
done > $O/upload_ab.txt 2>&1
cat $O/upload_ab.txt
