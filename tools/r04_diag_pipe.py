import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/ai-video-detector_amd')
import avd_hip
from avd_hip import synth
from oracle import oracle as O
src=open('/root/repo/tests/test_gpu_fbfast.py').read()
ns={}; exec(src[src.index("def _hard_frames"):src.index("def _check_pairs")], {"np":np}, ns)
frames=ns["_hard_frames"]()
with avd_hip.Context(0) as c:
    c.set_option("fb_mode",1); c.set_option("fb_rerun",0)
    c.set_option("fb_fold_up",0)
    c.farneback_pairs(frames)
    ref=[c.debug_fetch(f"flow{k}",(len(frames)-1,2,320>>k,320>>k),np.float32) for k in range(4)]
    c.set_option("fb_fold_up",8)
    c.farneback_pairs(frames)
    got=[c.debug_fetch(f"flow{k}",(len(frames)-1,2,320>>k,320>>k),np.float32) for k in range(4)]
for k in (3,2,1,0):
    d=np.abs(got[k].astype(np.float64)-ref[k])
    nd=(got[k].view(np.uint32)!=ref[k].view(np.uint32))
    print("level",k,"size",320>>k,"differing",int(nd.sum()),"max",d.max())
    if nd.any():
        p,cc,ys,xs=np.nonzero(nd)
        print("  pairs",np.unique(p),"comp",np.unique(cc),"rows",ys.min(),ys.max(),"cols",xs.min(),xs.max())
        # histogram by column and row (pair 0)
        for pp in np.unique(p): print("   pair",pp,"differing",int(nd[pp].sum()), "max", np.abs(got[k][pp].astype(np.float64)-ref[k][pp]).max(), "nan got", int(np.isnan(got[k][pp]).sum()), "nan ref", int(np.isnan(ref[k][pp]).sum()))
        m=nd[p[0]].any(axis=0)
        print("  rows with diffs (pair %d):"%p[0], np.nonzero(m.any(axis=1))[0][:40])
        print("  cols with diffs:", np.nonzero(m.any(axis=0))[0][:80])
        y,x=np.nonzero(m); print("  sample", [(int(y[i]),int(x[i]),float(got[k][p[0],0,y[i],x[i]]),float(ref[k][p[0],0,y[i],x[i]])) for i in range(0,len(y),max(1,len(y)//8))][:8])
