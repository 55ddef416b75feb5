import torch, time
dev=torch.device('cuda',0)
n=12*1024*1024
d=torch.empty(n,dtype=torch.uint8,device=dev)
h=torch.empty(n,dtype=torch.uint8).pin_memory()
def t(fn,it=50):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/it
print("D2H 12MB pinned GB/s", n/t(lambda: h.copy_(d,non_blocking=True))/1e9)
print("H2D 12MB pinned GB/s", n/t(lambda: d.copy_(h,non_blocking=True))/1e9)
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
h2=torch.empty(n,dtype=torch.uint8).pin_memory(); d2=torch.empty(n,dtype=torch.uint8,device=dev)
def two():
    with torch.cuda.stream(s1): h.copy_(d,non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2,non_blocking=True)
print("2x D2H concurrent streams GB/s", 2*n/t(two)/1e9)
def bidir():
    with torch.cuda.stream(s1): h.copy_(d,non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(h2,non_blocking=True)
print("D2H+H2D concurrent GB/s total", 2*n/t(bidir)/1e9)
hp=torch.empty(n,dtype=torch.uint8)
print("D2H pageable GB/s", n/t(lambda: hp.copy_(d),10)/1e9)
