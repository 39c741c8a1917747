import sys, time, importlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pkg=importlib.import_module("image-feature-extraction_amd"); synth=importlib.import_module("image-feature-extraction_amd.synthetic"); slab=importlib.import_module("image-feature-extraction_amd.slab")
class A: pass
for nz,spi,g in ((64,3,4),(64,1,3),(256,3,4)):
    a=A(); a.trig=2; a.i16=False; a.spacing=(1.0,1.0,1.0); a.line_groups=g; a.scales_per_item=spi
    r=slab.SlabRunner(pkg,synth,(nz,512,512),[1.0,2.0,4.0],3,"ones",pkg.INTERLEAVED,0,1,dev,a)
    for _ in range(3): r.step()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(20): r.step()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("nz",nz,"spi",spi,"g",g,"host enqueue ms/step %.3f  total ms/step %.3f"%((t1-t0)/20*1e3,(t2-t0)/20*1e3))
    r.finish(); del r
dist.destroy_process_group()
