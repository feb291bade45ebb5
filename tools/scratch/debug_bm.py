import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch, numpy as np, fixtures
from oracle import xfeat_oracle as O
from accelerated_features_amd import XFeat
sd=fixtures.synthetic_state_dict(0)
xf=XFeat(weights=sd, top_k=4096)
sa,sb=fixtures.star_pair(2,320,384,seed=41)
oa=O.detect_and_compute_dense(sd,sa,2048); ob=O.detect_and_compute_dense(sd,sb,2048)
bm=xf.batch_match(oa['descriptors'].cuda(), ob['descriptors'].cuda())
bo=O.batch_match(oa['descriptors'], ob['descriptors'])
for b in range(2):
    s=(oa['descriptors'][b].double()@ob['descriptors'][b].double().t())
    h0,h1=bm[b][0].cpu(),bm[b][1].cpu(); o0,o1=bo[b]
    print(b,'counts',len(h0),len(o0),'idx0 equal',torch.equal(h0,o0))
    hd=dict(zip(h0.tolist(),h1.tolist())); od=dict(zip(o0.tolist(),o1.tolist()))
    bad=[k for k in set(hd)|set(od) if hd.get(k)!=od.get(k)]
    print('  differing rows',len(bad))
    for k in bad[:10]:
        row=s[k]; t=torch.topk(row,3)
        print('   row',k,'hip',hd.get(k),'orc',od.get(k),'top3',t[1].tolist(),[round(v,5) for v in t[0].tolist()], 's[hip]',row[hd[k]].item() if k in hd else None)
    # same with single-pair match
    i0,i1=xf.match(oa['descriptors'][b].cuda(), ob['descriptors'][b].cuda(), min_cossim=-1)
    print('  match() equal to batch', torch.equal(i0.cpu(),h0), torch.equal(i1.cpu(),h1))
