import torch, sys
sys.path.insert(0,'/root/repo')
from digat_amd import training
g = torch.Generator().manual_seed(5)
V, dm = 5000, 400
table = torch.randn(V, dm, generator=g)
ids_a = torch.randint(0, 300, (320, 10), generator=g)
ids_b = torch.randint(100, V, (64, 50), generator=g)
ids_b[:, :5] = 7
wa, wb = torch.ones(320, 10, dm), torch.ones(64, 50, dm)
t1 = table.cuda().requires_grad_(True)
a, b = training.TableLookup2.apply(t1, ids_a.cuda(), ids_b.cuda())
((a * wa.cuda()).sum() + (b * wb.cuda()).sum()).backward()
torch.cuda.synchronize()
got = t1.grad[:, 0].cpu()
allids = torch.cat([ids_a.flatten(), ids_b.flatten()])
want = torch.bincount(allids, minlength=V).float()
bad = (got != want).nonzero().flatten()
print("bad ids", len(bad))
for i in bad[:10].tolist():
    pos = (allids == i).nonzero().flatten().tolist()
    print(i, "got", got[i].item(), "want", want[i].item(), "positions", pos[:12], "chunks", sorted(set(p // 64 for p in pos))[:12])
