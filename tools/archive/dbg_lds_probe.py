import ctypes, sys
import torch
lib = ctypes.CDLL("tools/liblds_probe.so")
lib.lds_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
sizes = [8448, 18688, 22784, 24832, 28928, 33024, 37120, 41216, 65536, 84224, 110848, 147712]


def probe(lds, via):
    out = torch.zeros(4, dtype=torch.int32, device=dev)
    lib.lds_probe(out.data_ptr(), 4, 512, lds, torch.cuda.current_stream().cuda_stream, via)
    return out


for via in (0, 1):
    for lds in sizes:
        e = probe(lds, via); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            probe(lds, via)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = probe(lds, via)
        g.replay(); torch.cuda.synchronize()
        ev, gv = int(e[0]) & 0xFFFFFFFF, int(o[0]) & 0xFFFFFFFF
        print(f"via_pointer={via} lds={lds:7d}  eager reg={ev:#010x} size_field={(ev >> 12) & 0x1FF}  graph reg={gv:#010x} size_field={(gv >> 12) & 0x1FF}", flush=True)
