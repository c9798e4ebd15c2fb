"""Where a wave of the image-path GEMM spends a tile's k-loop (a -DGG_STAMP_DMA build:
tools/variant.sh stamps grouped_gemm -DGG_STAMP_DMA; run with SPT_HIP_LIBRARY=.../libspt_hip_stamps.so
SPT_GEMM_PERSISTENT=0): cycles issuing the step's LDS-DMAs, reading fragments (+ the wait for them),
in the MFMA groups, parked at the step's barrier -- per tile, median over the waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, kk, dev = 1024, 1024, 'cuda'
torch.manual_seed(0)
for mt in [64, 512]:
    rows = 128 * mt
    extra = 2 * (mt + 1) * 8 * 4 * 8 * 8 // (d * 4) + 8
    one = torch.tensor([0, rows], dtype=torch.int32, device=dev)
    w = torch.randn([d, kk], device=dev); wi = ext.split_bf16(w)
    a = torch.randn([rows, kk], device=dev); ai = ext.split_bf16(a)
    big = torch.zeros([rows + extra, d], device=dev)
    call = lambda: ext.grouped_gemm_fused(a, w, one, 1, d, kk, 0, kk, 1, rows, a_image=ai, w_image=wi, out=big[:rows])
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    st = big[rows:].flatten().view(torch.int64)[:mt * 8 * 4 * 8].view(-1, 8).cpu().double()
    st = st[st[:, 4] > 0]
    med = st.median(0).values
    print('%d tiles: waves %d; per tile (cycles): dma issue %.0f | frag reads %.0f | mfma %.0f | barrier %.0f | loop %.0f'
          % (mt * 8, len(st), med[0], med[1], med[2], med[3], med[4]))
    print('   per k-step: dma %.0f reads %.0f mfma %.0f barrier %.0f total %.0f' % tuple((med[:5] / 32).tolist()))
    print('   outside the loop: entry -> loop %.0f | loop -> epilogue %.0f | epilogue %.0f' % (med[5], med[6], med[7]))
