import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from diffusers_amd import factory
DEV='cuda'
g = np.load('tests/golden/tiny_sdxl_pipeline.npz')
t = lambda k: torch.from_numpy(g[k]).to(torch.bfloat16).to(DEV)
pipe = factory.build_sdxl_pipeline(device=DEV, tiny=True, seed=0)
kw = dict(prompt_embeds=t("prompt_embeds"), negative_prompt_embeds=t("negative_prompt_embeds"),
          pooled_prompt_embeds=t("pooled"), negative_pooled_prompt_embeds=t("negative_pooled"),
          num_inference_steps=4, guidance_scale=5.0, height=128, width=128)
def S(m):
    torch.cuda.synchronize(); print(m, flush=True)
a = pipe(latents=t("latents").clone(), output_type="latent", use_graph=False, **kw).images.clone(); S('eager ok')
b = pipe(latents=t("latents").clone(), output_type="latent", use_graph=True, **kw).images.clone(); S('graph ok')
c = pipe(latents=t("latents").clone(), output_type="latent", use_graph=True, **kw).images.clone(); S('graph2 ok')
print(torch.equal(a,b), torch.equal(a,c))
lat = pipe(latents=t("latents").clone(), output_type="latent", use_graph=True, **kw).images; S('graph3 ok')
img = pipe.vae.decode(lat, return_dict=False, latents_div=float(pipe.vae.config.scaling_factor))[0]; S('decode ok')
img = pipe(latents=t("latents").clone(), output_type="raw", **kw).images; S('raw ok')
