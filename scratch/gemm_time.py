import sys, os, ctypes, torch, time
sys.path.insert(0, os.getcwd())
from proxytransformation_amd import _abi
lib=_abi.lib()
dev=torch.device('cuda')
def bench(R,N,K,gelu=0,res=False,reps=200, gap_us=0):
    x=torch.randn(R,K,device=dev); w=torch.randn(N,K,device=dev)*0.05; b=torch.randn(N,device=dev); y=torch.empty(R,N,device=dev)
    r=torch.randn(R,N,device=dev) if res else None
    st=torch.cuda.current_stream().cuda_stream
    def call(): 
        rc=lib.ptx_linear(x.data_ptr(),w.data_ptr(),b.data_ptr(),r.data_ptr() if res else None,y.data_ptr(),R,N,K,gelu,st); assert rc==0
    for _ in range(5): call()
    torch.cuda.synchronize()
    ref=torch.nn.functional.linear(x,w,b)
    if gelu: ref=torch.nn.functional.gelu(ref)
    if res: ref=ref+r
    err=(y-ref).abs().max().item()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    if gap_us==0:
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/reps*1e3
    else:
        tot=0
        for _ in range(50):
            torch.cuda.synchronize(); time.sleep(gap_us*1e-6)
            e0.record(); call(); e1.record(); torch.cuda.synchronize(); tot+=e0.elapsed_time(e1)
        t=tot/50*1e3
    fl=2.0*R*N*K
    print(f"R={R} N={N} K={K} gelu={gelu} gap={gap_us}us: {t:7.1f} us  {fl/t/1e6:7.1f} TF  err={err:.2e}")
for shp in [(1024,256,1024),(1024,1024,256),(1024,768,256),(1024,256,256),(784,768,512),(784,256,256),(2048,256,1024),(8192,256,1024)]:
    bench(*shp)
bench(1024,256,1024,gap_us=300)
bench(1024,1024,256,gap_us=300)
