import sys, numpy as np, torch
sys.path.insert(0, '.')
from diffusers_amd import ops, schedulers as S
from oracle.samplers import EulerOracle, cfg_combine
gz = np.load('tests/golden/schedulers.npz')
dev = torch.device('cuda')
def nb(a, b):
    a = a.float().contiguous(); b = b.float().contiguous()
    return int((a.view(torch.int32) != b.view(torch.int32)).sum())
for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    x0 = torch.from_numpy(gz[f'x0_{dt_name}']).to(dt).to(dev); eps = torch.from_numpy(gz[f'eps_{dt_name}']).to(dt).to(dev)
    e = S.EulerDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, timestep_spacing="leading")
    e.set_timesteps(5, device=dev)
    o = EulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
    o.set_timesteps(5)
    print(dt_name, 'sigmas ours==oracle', torch.equal(e.sigmas, o.sigmas), e.sigmas.numpy().view(np.int32), o.sigmas.numpy().view(np.int32))
    x = ops.mul_scalar(x0, float(e.init_noise_sigma))
    rs = torch.from_numpy(gz[f'euler_scaled_{dt_name}']); rt = torch.from_numpy(gz[f'euler_traj_{dt_name}'])
    for i, t in enumerate(e.timesteps):
        xin = x.cpu()
        sc = e.scale_model_input(x, t).cpu()
        sig = e.sigmas[i]
        sc_ref = xin / ((sig ** 2 + 1) ** 0.5)
        x = e.step(eps[i], t, x).prev_sample
        xo = x.cpu()
        s32 = xin.float(); mo = eps[i].cpu()
        po = s32 - sig * mo; der = (s32 - po) / sig; dtt = e.sigmas[i + 1] - sig; prev = (s32 + der * dtt).to(dt)
        print(dt_name, i, 'scaled vs local', nb(sc, sc_ref), 'vs golden', nb(sc, rs[i]), '| step vs local', nb(xo, prev), 'vs golden', nb(xo, rt[i]))
    # fused cfg
    u, c = eps[0].cpu(), eps[1].cpu()
    comb = cfg_combine(u, c, 7.5)
    o.set_timesteps(5)
    want = o.step(comb, x0.cpu())
    e.set_timesteps(5, device=dev)
    got = e.step_cfg(torch.cat([u, c], 0).to(dev), x0, 7.5).cpu()
    e.set_timesteps(5, device=dev)
    got2 = e.step(comb.to(dev), e.timesteps[0], x0).prev_sample.cpu()
    print(dt_name, 'cfg fused vs oracle', nb(got, want), 'unfused kernel vs oracle', nb(got2, want), 'fused vs unfused', nb(got, got2))
