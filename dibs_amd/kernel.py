"""Kernel descriptors of the SVGD update.  The engine evaluates them on the device (k_kmat in
dibs_amd/csrc/kernels_marginal.h); these classes carry the hyper-parameters and offer a host ``eval`` for
inspection.  Reference: dibs/kernel.py:4-30 (AdditiveFrobeniusSEKernel), :33-71 (JointAdditive...)."""
import numpy as np


class AdditiveFrobeniusSEKernel:
    def __init__(self, *, h=20.0, scale=1.0):
        self.h = h
        self.scale = scale

    def eval(self, *, x, y):
        return self.scale * np.exp(-np.sum((np.asarray(x) - np.asarray(y)) ** 2.0) / self.h)


class JointAdditiveFrobeniusSEKernel:
    def __init__(self, *, h_latent=5.0, h_theta=500.0, scale_latent=1.0, scale_theta=1.0):
        self.h_latent = h_latent
        self.h_theta = h_theta
        self.scale_latent = scale_latent
        self.scale_theta = scale_theta

    def eval(self, *, x_latent, x_theta, y_latent, y_theta):
        from .utils.tree import tree_leaves
        zn = np.sum((np.asarray(x_latent) - np.asarray(y_latent)) ** 2.0)
        tn = sum(np.sum((a - b) ** 2.0) for a, b in zip(tree_leaves(x_theta), tree_leaves(y_theta)))
        return self.scale_latent * np.exp(-zn / self.h_latent) + self.scale_theta * np.exp(-tn / self.h_theta)
