import torch, time
dev='cuda:0'
def t(fn, reps=50):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for mb in (16.8, 67, 268, 1072):
    n=int(mb*1e6/4)
    pool=[torch.randn(n,device=dev) for _ in range(max(2,int(600/mb)))]  # rotate buffers to defeat the 256 MB cache
    out=torch.empty(n,device=dev)
    i=[0]
    def cp():
        i[0]=(i[0]+1)%len(pool); out.copy_(pool[i[0]])
    def sm():
        i[0]=(i[0]+1)%len(pool); return pool[i[0]].sum()
    us=t(cp); print(f"copy {mb} MB: {us:.1f} us  {2*mb/us*1e0:.2f} TB/s (r+w)")
    us=t(sm); print(f"sum  {mb} MB: {us:.1f} us  {mb/us:.2f} TB/s (read)")
