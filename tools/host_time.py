"""Host enqueue time against GPU time of a bench.py step (is the step launch-bound anywhere?).
   python tools/host_time.py [workload] [graph] [samples]   (bert_base_train by default; `graph`: the training step through
   training.GraphedTrainingStep — two eager steps, then replays; otherwise the step is not graphed)
Prints, per step: the time the host needs to enqueue the step (no sync inside) and the synchronised step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import bayeformers_amd as bf

w = sys.argv[1] if len(sys.argv) > 1 else "bert_base_train"
device = torch.device("cuda:0")
S, dtype = bench.DEFAULTS[w][0], bench.DEFAULTS[w][1]
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
if len(sys.argv) > 3:
    S = int(sys.argv[3])
bf.set_compute_dtype(dtype)
bf.manual_seed(0x5EED)
if w.startswith("bert_base"):
    step = bench.make_bert(device, S, dtype, train=w.endswith("train"), train_mode=True, graph_train=graph)[0]
else:
    step = bench.make_bert_large_qa(device, S, dtype, train=w.endswith("train"), train_mode=True, graph_train=graph)[0]
for _ in range(4):
    step()
torch.cuda.synchronize()
host, n = [], 12
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    step()
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n
# one isolated step: host and GPU start together
torch.cuda.synchronize()
a = time.perf_counter(); step(); h1 = time.perf_counter() - a; torch.cuda.synchronize(); g1 = time.perf_counter() - a
print(f"{w} S={S} graph={graph and bench._train_graph_note(step)}: host enqueue per step (pipelined loop) {[round(x * 1e3, 1) for x in host]} ms; step time {total * 1e3:.2f} ms; "
      f"isolated step: host {h1 * 1e3:.2f} ms, until GPU done {g1 * 1e3:.2f} ms")
