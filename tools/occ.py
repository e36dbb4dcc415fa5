import ctypes as C, os, sys, torch
torch.cuda.init(); torch.zeros(1, device='cuda')
h = C.CDLL(os.environ.get('GADAPT_LIB', 'g_adaptivity_amd/libgadapt_hip.so'))
for c in (8, 32, 64, 128):
    out = (C.c_int * 3)()
    h.gadapt_debug_occupancy(c, out)
    print("C", c, "blocks/CU per runtime: fwd", out[0], "bwd_target", out[1], "bwd_source", out[2])
p = torch.cuda.get_device_properties(0)
print(p.name, p.multi_processor_count, "shared/block", p.shared_memory_per_block, "shared/mp", getattr(p, 'shared_memory_per_multiprocessor', None), "regs/mp", p.regs_per_multiprocessor)
