import ctypes as C, os, sys
order = sys.argv[1]
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l))
def mine():
    lib = C.CDLL('/root/repo/slam-sdvl_amd/csrc/libsdvl_hip.so')
    h = C.c_void_p()
    rc = lib.sdvl_ctx_create(0, C.byref(h))
    print('sdvl_ctx_create rc', rc)
    hip = C.CDLL('libamdhip64.so.7')
    n = C.c_int(-1)
    e = hip.hipGetDeviceCount(C.byref(n))
    hip.hipGetErrorString.restype = C.c_char_p
    print('hipGetDeviceCount', e, n.value, hip.hipGetErrorString(e))
def tor():
    import torch
    print('torch avail', torch.cuda.is_available(), torch.version.hip)
    if order.endswith('init'):
        x = torch.zeros(4, device='cuda'); print('torch tensor ok', x.sum().item())
if order.startswith('torch'):
    tor(); print(maps()); mine(); print(maps())
else:
    mine(); print(maps()); tor(); print(maps())
    import torch
    x = torch.ones(4, device='cuda'); print('torch after mine ok', x.sum().item())
