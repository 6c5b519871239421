import sys, time, os, ctypes as C
sys.path.insert(0,'/root/repo')
os.environ.setdefault("GPU_MAX_HW_QUEUES","12")
import torch, numpy as np
from __graft_entry__ import load_package
pkg=load_package()
ch=pkg.synth_descriptors(1,nch=12,seed=5)[0]
nsamp=300000; delt=1/2.6e6
L=pkg.lib()
def run(name, ptr, n=6000, opt=None):
    with pkg.Synth(0) as s:
        if opt: s.set_option(*opt)
        st=np.zeros(12,pkg.STATE_DTYPE)
        lat=[]
        for k in range(n):
            t=time.perf_counter()
            rc=L.gpsbb_fill_block_ex(s._h, ch.ctypes.data, 12, delt, nsamp, 0, ptr, st.ctypes.data)
            lat.append((time.perf_counter()-t)*1e3)
            assert rc==0
        a=np.array(lat[10:])
        slow=[(i+10,round(x,2)) for i,x in enumerate(a) if x>2.0]
        print(name, "p50 %.3f p99 %.3f p999 %.3f max %.3f n>2ms %d"%(np.percentile(a,50),np.percentile(a,99),np.percentile(a,99.9),a.max(),len(slow)), slow[:12], flush=True)
page=np.zeros((nsamp,2),np.int16)
pin=torch.empty((nsamp,2),dtype=torch.int16).pin_memory()
run("pageable", page.ctypes.data)
run("pinned  ", pin.data_ptr())
run("pageable, device pre-pass", page.ctypes.data, opt=(pkg.OPT_SEED_WHERE,1))
run("pinned, device pre-pass", pin.data_ptr(), opt=(pkg.OPT_SEED_WHERE,1))
