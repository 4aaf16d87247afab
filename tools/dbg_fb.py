import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, gstpeaq_amd
import cases as case_defs, oracle_lib as orc
ctx=gstpeaq_amd.Context(0)
ref,test=case_defs.make_inputs(dict(kind="synth", seed=5, channels=1, n=30000))
nb=30000//192
got=gstpeaq_amd.debug_filterbank(ctx, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(), nb, 320)
exp=orc.fbear(np.ascontiguousarray(ref[:,0]), nb)
rel=np.abs(got[:,0,0:40]-exp["unsmeared"])/np.abs(exp["unsmeared"])
bad=np.argwhere(rel>1e-9)
print("bad blocks:", sorted(set(bad[:,0].tolist())))
print("bad bands:", sorted(set(bad[:,1].tolist())))
for b in sorted(set(bad[:,0].tolist()))[:6]:
    print(b, np.flatnonzero(rel[b]>1e-9).tolist(), rel[b].max())
