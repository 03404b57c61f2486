"""Build-container measurement for bench.py's cpu_baseline: the oracle port (oracle/ggnn_oracle.py, what the GPU box
can run) against the UNMODIFIED reference model (/root/reference/graphinvent/gnn/mpnn.py, which cannot travel) on the
same host cores, same weights, same B = 1000 headline batch, same training step (fwd + KL + bwd + Adam).
Writes profiles/<round>/port_over_reference.json; bench.py quotes the ratio next to a "port" baseline.
    python tools/port_over_reference.py [round_dir] [threads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from graphinvent_amd import synthetic
from oracle import ggnn_oracle as O

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(8, os.cpu_count() or 1)
ref_root = os.environ.get("GI_REFERENCE", "/root/reference")
torch.set_num_threads(threads)
cfg, _ = bench.workload_constants("cpu")
ocfg = {k: cfg[k] for k in O.GDB13_DEFAULTS}; ocfg["device"] = "cpu"
sh = synthetic.SHAPES["gdb13"]
n8, e8, a8 = synthetic.make_batch(bench.BATCH, **sh, seed=0)
nodes, edges, tgt = (torch.from_numpy(x).float() for x in (n8, e8, a8))


def time_model(model, n=3):
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    def one():
        out = model(nodes, edges); opt.zero_grad(); loss = O.kl_loss(out, tgt); loss.backward(); opt.step(); return float(loss)
    one()
    t0 = time.perf_counter()
    for _ in range(n): last = one()
    return (time.perf_counter() - t0) / n, last

port = O.OracleGGNN(ocfg, seed=0)
sys.path.insert(0, os.path.join(ref_root, "graphinvent"))
import gnn.mpnn as ref_mpnn                                   # the unmodified reference package
ref = ref_mpnn.GGNN(O.as_constants(ocfg))
ref.load_state_dict({k: v.detach().clone() for k, v in port.named_oracle_params().items()})
t_ref, l_ref = time_model(ref)
t_port, l_port = time_model(port)
out = {"batch": bench.BATCH, "threads": threads, "host_cpus": os.cpu_count(),
       "reference_graphs_per_s": round(bench.BATCH / t_ref, 1), "port_graphs_per_s": round(bench.BATCH / t_port, 1),
       "port_over_reference": round(t_ref / t_port, 3), "loss_after_4_steps": [round(l_ref, 5), round(l_port, 5)],
       "torch": torch.__version__, "reference": os.path.join(ref_root, "graphinvent/gnn/mpnn.py"),
       "note": "same weights (state_dict of the port loaded into the reference model), same batch, 3 timed steps after 1 warm-up each"}
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
with open(os.path.join(ROOT, "profiles", rnd, "port_over_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
