"""CPU restatement of the two callers of the network.  TEST INFRASTRUCTURE.

  * train_step       -- /root/reference/train.py:127-140 (zero_grad, forward, CE(ignore 255, mean),
                        backward, Adam.step) on the oracle's torchsparse restatement.
  * inference_post   -- /root/reference/score/prob_inference.py:100-113 (voxel->point gather by
                        inverse indices, softmax, view mean, argmax).
"""
import numpy as np
import torch

from oracle import tsref


def forward(model, feats, coords):
    return model(tsref.SparseTensor(feats, coords))


def train_step(model, optimizer, feats, coords, labels):
    optimizer.zero_grad()
    logits, _ = forward(model, feats, coords)
    loss = torch.nn.functional.cross_entropy(logits, labels, ignore_index=255, reduction='mean')
    loss.backward()
    optimizer.step()
    return loss.detach(), logits.detach()


def inference_post(logits_v_b, inverse_indices_b, inf_reps):
    logits_p_b = logits_v_b.cpu()[inverse_indices_b]
    prob_map = torch.softmax(logits_p_b, dim=1)
    prob_map = prob_map.numpy().reshape(inf_reps, -1, prob_map.shape[-1])
    prob_map_mean = np.mean(prob_map, axis=0)
    pred = np.argmax(prob_map_mean, axis=1)
    return prob_map_mean, pred
