import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sparsebase_amd import ops, synth
rp, col = synth.grid_graph(1024, 1024)
drp, dcol = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
out = torch.empty(len(rp) - 1, dtype=torch.int32, device="cuda")
ops.rcm_reorder(drp, dcol, out=out); torch.cuda.synchronize()
ops.rcm_reorder(drp, dcol, out=out); torch.cuda.synchronize()
