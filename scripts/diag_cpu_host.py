"""Why did the driver's un-tasksetted `pytest -m gpu` spend minutes in a CPU pass that takes seconds under `taskset -c 0-7`?
Times one OPT-350m-shaped teacher-forced pass (24 layers, S rows, torch CPU fp32) in fresh subprocesses under several host
conditions and prints the cgroup's throttling counters around each.  Report only; output -> stdout."""
import os
import subprocess
import sys
import time

CHILD = r"""
import os, sys, time
thr = int(sys.argv[1]); gpu = int(sys.argv[2]); S = int(sys.argv[3])
for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = str(thr)
import torch
torch.set_num_threads(thr)
if gpu:
    torch.cuda.init(); a = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
g = torch.Generator().manual_seed(0)
H, F, L = 1024, 4096, 24
W = [torch.randn(H, H, generator=g) * 0.02 for _ in range(4)] + [torch.randn(F, H, generator=g) * 0.02, torch.randn(H, F, generator=g) * 0.02]
h = torch.randn(1, S, H, generator=g)
def layer(h):
    q, k, v = [(h @ W[i].t()).view(1, S, 16, 64).permute(0, 2, 1, 3) for i in range(3)]
    w = (q @ k.transpose(-1, -2)) * 0.125
    w = w.masked_fill(torch.ones(S, S, dtype=torch.bool).triu(1), float("-inf"))
    a = (torch.softmax(w, -1) @ v).permute(0, 2, 1, 3).reshape(1, S, H)
    h = torch.nn.functional.layer_norm(h + a @ W[3].t(), (H,))
    return torch.nn.functional.layer_norm(h + torch.relu(h @ W[4].t()) @ W[5].t(), (H,))
layer(h)
t0 = time.time()
for _ in range(L):
    h = layer(h)
dt = time.time() - t0
print(f"threads={thr} gpu_ctx={gpu} S={S} affinity={len(os.sched_getaffinity(0))}: {dt:.2f} s for {L} layers", flush=True)
"""


def cg():
    out = {}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            with open(f) as fh:
                out[f] = " ".join(fh.read().split())
        except OSError:
            pass
    return out


def main():
    S = int(os.environ.get("DIAG_S", "1757"))
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", open("/proc/loadavg").read().strip())
    print("cgroup", cg())
    runs = [([], 8, 0), ([], 8, 1), (["taskset", "-c", "0-7"], 8, 1), ([], 1, 1), (["taskset", "-c", "0"], 1, 1), ([], 32, 1)]
    for pre, thr, gpu in runs:
        t0 = time.time()
        try:
            r = subprocess.run(pre + [sys.executable, "-c", CHILD, str(thr), str(gpu), str(S)], capture_output=True, text=True, timeout=400)
            line = (r.stdout.strip().splitlines() or ["<no output> " + r.stderr[-300:]])[-1]
        except subprocess.TimeoutExpired:
            line = "TIMEOUT 400 s"
        print(f"[{' '.join(pre) or 'no taskset'}] {line}  (process wall {time.time() - t0:.1f} s)", flush=True)
        print("   cgroup", cg().get("/sys/fs/cgroup/cpu.stat", ""), flush=True)


if __name__ == "__main__":
    main()
