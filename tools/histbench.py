import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from image_classification_amd import ops
from image_classification_amd.synthetic import image_offsets
dev="cuda"
g=torch.Generator(device=dev); g.manual_seed(1)
for n_img, per, k, norm in ((1000,1000,10000,"l2"),(1000,1000,10000,"l1"),(2000,2000,10000,"power_l2"),(100,1000,1000,"l1")):
    codes=torch.randint(0,k,(n_img*per,),device=dev,generator=g,dtype=torch.int32)
    off=torch.from_numpy(image_offsets(n_img,per)).to(dev)
    for _ in range(3): ops.hist(codes,off,k,norm)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): h=ops.hist(codes,off,k,norm)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print("hist %d x %d codes, %d bins, %s: %.1f us  (%.2f TB/s of output)"%(n_img,per,k,norm,ms*1e3, n_img*k*8/ms/1e9))
