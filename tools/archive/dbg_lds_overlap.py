import ctypes
import torch
lib = ctypes.CDLL("tools/liblds_probe.so")
lib.lds_pattern.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
for sa, ta in ((20640, 256), (20736, 256), (65536, 256)):
    for sb, tb in ((37120, 512), (110848, 512), (84224, 512), (28928, 512), (18688, 512), (147712, 512)):
        err = torch.zeros(2, dtype=torch.int32, device=dev)
        for it in range(10):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                lib.lds_pattern(err.data_ptr(), 1, 4096, ta, sa, 8, side.cuda_stream)
            lib.lds_pattern(err.data_ptr() + 4, 2, 1024, tb, sb, 8, torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        print(f"A: {sa:6d} B x {ta} thr   B: {sb:6d} B x {tb} thr   errors A {int(err[0])}  B {int(err[1])}", flush=True)
