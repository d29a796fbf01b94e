import torch, time, os, sys
dev="cuda:0"
def run(tag):
    torch.manual_seed(0)
    head = torch.nn.Sequential(torch.nn.Linear(256,128), torch.nn.ReLU(), torch.nn.Linear(128,1024), torch.nn.ReLU(), torch.nn.Linear(1024,1024), torch.nn.ReLU(), torch.nn.Linear(1024,512), torch.nn.ReLU(), torch.nn.Linear(512,1)).to(dev)
    x = torch.randn(512,256,device=dev, requires_grad=True)
    for _ in range(10):
        head(x).sum().backward()
    torch.cuda.synchronize()
    a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        head.zero_grad(); head(x).sum().backward()
    b.record(); torch.cuda.synchronize()
    print(tag, a.elapsed_time(b)/50*1000, "us per fwd+bwd")
run("default")
try:
    torch.backends.cuda.preferred_blas_library("hipblaslt"); run("hipblaslt")
except Exception as e: print("hipblaslt pref failed", e)
try:
    torch.backends.cuda.preferred_blas_library("cublas"); run("rocblas")
except Exception as e: print("rocblas pref failed", e)
