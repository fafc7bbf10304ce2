import importlib, sys, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import orc
ebo=importlib.import_module('event-based-odomety_amd'); synth=importlib.import_module('event-based-odomety_amd.synth')
for w in (109,139):
    ev,_=synth.make_window(0,window=w,n_events=15000)
    prm=orc.default_params(loss=0); prmf=orc.default_params(loss=0,tv_weight=0.0)
    fo,_,so=orc.compensate_events_contrast(ev,prm,orc.default_solver(),want_image=False)
    with ebo.Context(image_w=240,image_h=180,patch_w=20,patch_h=20,loss=ebo.LOSS_EDGE,max_events=len(ev)) as c:
        c.set_window(ev)
        fh,sh=c.solve(ebo.default_solver())
        for it in (1,2,3,4,6,8,10,12,14):
            fi,si=c.solve(ebo.default_solver(max_num_iterations=it))
            foi,_,soi=orc.compensate_events_contrast(ev,prm,orc.default_solver(max_num_iterations=it),want_image=False)
            # J at the oracle's iterate
            r,J=c.eval(foi); ro,Jo,act,_=orc.window_eval(ev,prmf,foi)
            dJ=np.abs(J[0]-Jo); rel=dJ/(np.abs(Jo)+1e-12)
            print(w,'it',it,'dflow %.2e'%np.abs(fi[0]-foi).max(),'iters',si[0].iterations,soi.iterations,'| at oracle iterate: max|dr| %.2e max|dJ| %.2e maxrel %.2e worst patch %d'%(np.abs(r[0]-ro).max(), dJ.max(), rel.max(), int(rel.max(axis=1).argmax())))
