import sys, os, time, json, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
from gnn_pretraining_amd import synthetic as S, operators as O
from gnn_pretraining_amd.graph import Batch
from gnn_pretraining_amd.models import FinetuneGNN
dev = torch.device("cuda:0")
torch.set_num_threads(1)
gen = torch.Generator().manual_seed(0)
torch.manual_seed(0)
m = FinetuneGNN(dev, "Cora_NC", "full_finetune"); m.train()
g = S.cora_like(gen)
data = Batch.from_data_list([g]).to(dev)
idx = torch.randperm(2708, generator=gen)[:140].to(dev); y = g.y.to(dev)[idx]
opt = torch.optim.AdamW(m.param_groups)
def step():
    loss = O.cross_entropy_sum(O.take_rows(m(data), idx), y) / 140
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("ms/step", round((time.perf_counter() - t0) / 50 * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
