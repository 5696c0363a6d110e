import sys, torch
sys.path.insert(0, '.')
from diffusers_amd import ops, _lib as L
dev = torch.device('cuda')
for (M, K, N) in [(32, 64, 64), (32, 64, 128), (64, 64, 64), (32, 128, 64), (256, 64, 64)]:
    for tile in (L.TILE_AUTO, L.TILE_64x64, L.TILE_128x64, L.TILE_64x128, L.TILE_128x128):
        for st in (L.STAGE_REGISTER, L.STAGE_LDS_DIRECT):
            x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
            try:
                y = ops.linear(x, w, tile=tile, staging=st)
                torch.cuda.synchronize()
                err = (y.float() - x.float() @ w.float().t()).abs().max().item()
                print(M, K, N, tile, st, 'ok', err)
            except Exception as e:
                print(M, K, N, tile, st, 'FAIL', e)
