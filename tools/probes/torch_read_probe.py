import torch, time
n=1<<30
src=torch.empty(n//4,dtype=torch.int32,device='cuda').fill_(1)
f32=src.view(torch.float32)
def t(name,fn,moved,reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, round(moved*reps/(e0.elapsed_time(e1)*1e-3)/1e9,1))
t('sum_i32',lambda: src.sum(),n)
t('sum_f32',lambda: f32.sum(),n)
t('max_i32',lambda: src.max(),n)
t('amax_f32',lambda: f32.amax(),n)
t('sum_i64view',lambda: src.view(torch.int64).sum(),n)
t('any',lambda: src.view(torch.uint8).any(),n)
t('count_nonzero',lambda: torch.count_nonzero(src),n)
t('dot',lambda: torch.dot(f32,f32),2*n)
