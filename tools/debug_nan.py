import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops
from dp_cgans_amd.data_sampler import DataSampler
from dp_cgans_amd.engine import StepEngine
from dp_cgans_amd.synthetic import build_encoded_table
dev = torch.device('cuda:0')
table, span = build_encoded_table('T1M', dev, n_rows=int(sys.argv[1]))
print('table nan', torch.isnan(table).any().item(), table.shape, table[:2, :12])
s = DataSampler(table, span, device=dev)
print('cnt diag', s.cnt.diag()[:10])
eng = StepEngine(span, 128, (256, 256), (256, 256), 8192, 8, dev, private=True, seed=0)
eng.sampler = s
def rep(tag):
    torch.cuda.synchronize()
    items = dict(Hg=eng.Hg, fake=eng.fake, Xd=eng.Xd, y=eng.y, dg=eng.disc.g, dp=eng.disc.p, gg=eng.gen.g, gp=eng.gen.p,
                 H0=eng.H[0], S0=eng.S[0], D0=eng.Dl[0], Z0=eng.Z[0], dfake=eng.dfake, dX=eng.dX, dHg=eng.dHg)
    bad = [k for k, v in items.items() if not torch.isfinite(v).all()]
    print(tag, 'losses', eng.losses.cpu().tolist(), 'rowid min/max', eng.row_id.min().item(), eng.row_id.max().item(), 'nonfinite:', bad)
def bad():
    torch.cuda.synchronize()
    return not torch.isfinite(eng.losses).all().item()
for it in range(int(sys.argv[2])):
    eng.d_step()
    if bad(): rep('D%d' % it); break
    eng.g_step()
    if bad(): rep('G%d' % it); break
    if it % 10 == 0: rep('ok%d' % it)
