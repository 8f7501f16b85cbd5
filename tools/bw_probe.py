import torch, time
x=torch.empty((1<<28,),dtype=torch.float32,device='cuda')   # 1 GiB
y=torch.empty_like(x)
for name,fn in (("fill 1GiB",lambda: x.fill_(1.0)),("copy 1GiB->1GiB",lambda: y.copy_(x))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(name, "%.3f ms"%(dt*1e3), "%.2f TB/s written"%(x.numel()*4/dt/1e12))
