"""GPU box: what configs[3] would run at if the kNN launches cost nothing - the same bench loop (4 rooms per launch, N launches in
flight, 50 PGD iterations) with the graphs of a first forward teacher-forced (psg_gcn_set_graphs: no kNN kernel is launched).
Results are those of another network (fixed graphs); timing only: the bound the non-kNN kernels put on the workload."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import gcn28_state_dict, make_rooms, rule_labels

DB, conc, n_launch = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 6
model = runtime.GCNModel(gcn28_state_dict(), 28)
wss = [runtime.GCNWorkspace(DB, 4096, 28) for _ in range(conc)]
streams = [torch.cuda.Stream() for _ in range(conc)]
rooms = [make_rooms(DB, 5000 + s) for s in range(n_launch + conc)]
imgs = [torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).cuda() for r in rooms]
labs = [torch.from_numpy(rule_labels(r).astype(np.int32)).cuda() for r in rooms]
outs = [torch.empty_like(x) for x in imgs]
for mode in ("dynamic graphs (the bench)", "graphs teacher-forced: no kNN launches"):
    if mode.startswith("graphs"):
        for w in wss:
            w.forward(model, torch.from_numpy(np.ascontiguousarray(rooms[0])).cuda())
            nbr = torch.stack([w.edges(e) for e in range(28)]).contiguous()
            w.set_graphs(nbr)
    def launch(i):
        with torch.cuda.stream(streams[i % conc]):
            wss[i % conc].nb_attack(model, imgs[i], labs[i], 0.3, 2 / 255, 50, out=outs[i])
    for i in range(conc):
        launch(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(conc, conc + n_launch):
        launch(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-45s %d in flight: %.2f rooms/s, %.3f ms per 4-room iteration" % (mode, conc, DB * n_launch / dt, dt / n_launch / 50 * 1e3), flush=True)
