#!/bin/bash
# round 4: what does the GPU box offer for a whole-genome sized file? (disk, RAM, CPUs, zlib speed) + baseline suite
mkdir -p gpurun_out/r4
{
df -h / /tmp /dev/shm /root 2>&1
free -g
nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max
lscpu | head -20
rocm-smi --showmeminfo vram 2>&1 | head
python - <<'P'
import zlib, time, numpy as np
rng=np.random.default_rng(1)
# a BAM-like block: 36 B header-ish + name + 75 B random nibbles + 150 B binned quals
recs=[]
for i in range(220):
    recs.append(rng.integers(0,256,36,dtype=np.uint8).tobytes()+(b"q%09d\0"%i)+rng.integers(0,256,75,dtype=np.uint8).tobytes()+rng.choice(np.array([2,12,23,37],np.uint8),150,p=[.03,.07,.15,.75]).tobytes())
raw=b"".join(recs)[:65280]
for lvl in (1,4,6):
    t=time.time(); n=0
    while time.time()-t<1.0:
        c=zlib.compressobj(lvl,zlib.DEFLATED,-15); o=c.compress(raw)+c.flush(); n+=1
    dt=time.time()-t
    print("zlib level",lvl,"MB/s in",len(raw)*n/dt/1e6,"ratio",len(o)/len(raw))
P
dd if=/dev/zero of=/tmp/ddtest bs=1M count=4096 2>&1 | tail -1; rm -f /tmp/ddtest
} > gpurun_out/r4/probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputest0.txt 2>&1
tail -3 gpurun_out/r4/gputest0.txt
cat gpurun_out/r4/probe.txt
