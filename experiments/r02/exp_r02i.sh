mkdir -p gpurun_out/r02i
cat > /tmp/kb.py <<'PY'
import ctypes as C, sys, os, torch, time
sys.path.insert(0, os.getcwd())
from longlive_amd import _lib, ops
lib=_lib.load()
def run(tag, flags):
    _lib.check(lib.ll_set_tuning(b"gemm_lds_epi", flags), "t")
    out=[]
    for name,(M,N,K,epi) in {"o":(4680,1536,1536,2),"f2":(4680,1536,8960,2),"cq":(4680,1536,1536,0)}.items():
        x=torch.randn(M,K,device="cuda").bfloat16(); w=(torch.randn(N,K,device="cuda")/K**0.5).bfloat16(); b=torch.zeros(N,device="cuda").bfloat16()
        res=torch.randn(M,N,device="cuda").bfloat16(); e=torch.randn(1,3,6,N,device="cuda").bfloat16(); mod=torch.randn(6,N,device="cuda").bfloat16()
        kw=dict(res=res,e=e,mod=mod,gate_idx=2,rows_per_batch=M,frame_len=M//3) if epi==2 else {}
        for _ in range(3): ops.gemm(x,w,b,epi,out=res.clone() if epi==2 else None,**kw)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(30): ops.gemm(x,w,b,epi,**kw)
        torch.cuda.synchronize(); out.append(f"{name} {(time.perf_counter()-t0)/30*1e6:.1f}us")
    print(tag, " ".join(out), flush=True)
for rep in range(2):
    run("normal       ", 1)
    run("no-loop-DMA  ", 1|0x100)
    run("no-compute   ", 1|0x200)
    run("neither      ", 1|0x300)
PY
python /tmp/kb.py 2>&1 | grep -v amdgpu.ids
