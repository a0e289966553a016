import sys, os
sys.path[:0]=['.','tests']
order=sys.argv[1]
if order=='torch_first':
    import torch; print("torch avail", torch.cuda.is_available(), torch.version.hip)
    x=torch.zeros(10,device='cuda:0'); torch.cuda.synchronize()
import bzip3_amd
lib=bzip3_amd.load(); print("lib devices", lib.bz3_hip_device_count())
import datagen
d=datagen.shakespeare()[:200000]
st=bzip3_amd.State(1<<20, lib); r=st.encode_block(d); print("enc", r[0], r[1])
if order=='lib_first':
    import torch; print("torch avail", torch.cuda.is_available(), torch.version.hip)
import torch
cap=lib.bz3_bound(1<<20)+64
buf=torch.zeros(cap,dtype=torch.uint8,device='cuda:0'); buf[:len(d)]=torch.frombuffer(bytearray(d),dtype=torch.uint8).to('cuda:0'); torch.cuda.synchronize()
n=lib.bz3_hip_encode_block_device(st.ptr, buf.data_ptr(), len(d)); print("dev enc", n, bytes(buf[:n].cpu().numpy())==r[2])
os.system("cat /proc/%d/maps | grep -i amdhip | awk '{print $6}' | sort -u" % os.getpid())
