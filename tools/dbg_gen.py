import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import rel_err, load_golden
from gan_amd.modelling import build_generator
from oracle import gan_oracle as O
from oracle.detparams import det_state, det_uniform
from oracle.tape import RandTape
dev = torch.device('cuda:0')
g = load_golden('generator16')
shapes = O.stylegan2_generator_shapes(g['img_size'], base_dim=g['base_dim'])
st = det_state(shapes, seed=g['seed'])
gen = build_generator('stylegan2', img_size=g['img_size'], base_dim=g['base_dim']); gen.load_state_dict(st); gen.to(dev)
with RandTape(list(g['noise'])):
    img = gen(g['z'].to(dev))
loss = (img * det_uniform(tuple(img.shape), 902).to(dev)).sum() + img.square().mean()
loss.backward()
sd = {k: v.clone().double().requires_grad_(True) for k, v in st.items()}
img_o = O.generator(sd, g['z'].double(), [n.double() for n in g['noise']])
((img_o * det_uniform(tuple(img_o.shape), 902).double()).sum() + img_o.square().mean()).backward()
print('img', rel_err(img, img_o))
named = dict(gen.named_parameters())
for k, v in sd.items():
    print(f"{k:45s} {rel_err(named[k].grad, v.grad):.3e}")
