import torch, numpy as np
x = torch.zeros(64, device="cuda")
a = torch.randn(4096, 4096, device="cuda"); 
def pair(fn, n=200):
    out=[]
    for _ in range(n):
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1)*1e3)
    return np.median(out), np.min(out)
print("nothing between the events: median %.2f us min %.2f"%pair(lambda: None))
print("tiny kernel: median %.2f us min %.2f"%pair(lambda: x.add_(1)))
print("two tiny kernels: median %.2f us min %.2f"%pair(lambda: (x.add_(1), x.add_(1))))
print("4096^3 f32 matmul: median %.2f us min %.2f"%pair(lambda: torch.mm(a,a)))
def back2back(k, n=50):
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): torch.mm(a,a)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
print("4096^3 matmul back to back x50: %.2f us each"%back2back(0))
