import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch
from lidal_amd.network import SPVCNN, MinkUNet
from lidal_amd.train_step import forward_backward
from weights import fill_state_dict
g = np.load('/root/repo/tests/golden/model_small.npz')
def rel(a,b):
    a=np.asarray(a,np.float64); b=np.asarray(b,np.float64); return np.abs(a-b).max()/np.abs(b).max()
for name, cls in (('minkunet', MinkUNet), ('spvcnn', SPVCNN)):
    for rep in range(2):
        model = fill_state_dict(cls(19)).cuda().train()
        if hasattr(model, 'dropout'): model.dropout.p = 0.0
        loss, logits = forward_backward(model, torch.from_numpy(g['feats']).cuda(), torch.from_numpy(g['coords']).cuda(), torch.from_numpy(g['labels']).cuda())
        named = dict(model.named_parameters())
        norms = np.array([named[k].grad.norm().item() for k in g[name+'_grad_keys']])
        print(name, 'loss', loss.item(), float(g[name+'_train_loss']), 'logits', rel(logits.detach().cpu().numpy(), g[name+'_train_logits']))
        print('  norms-1', norms/g[name+'_grad_norms']-1)
        print('  stem', rel(named['stem.0.kernel'].grad.cpu().numpy(), g[name+'_grad_stem']), 'up1', rel(named['up1.0.net.0.kernel'].grad.cpu().numpy()[:, :8,:8], g[name+'_grad_up1dc']))
