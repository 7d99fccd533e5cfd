import sys; sys.path.insert(0,'/root/repo')
import hsamd; hs=hsamd.load(); L=hs._lib.lib()
for w in (1,2):
    for it in (200000, 2000000):
        print("waves/simd",w,"iters",it,"const-ish operands %.2f TF/s   random operands %.2f TF/s"%(L.hsk_mfma_f64_peak(w,it), L.hsk_mfma_f64_peak_random(w,it)), flush=True)
