"""Gaussian toy experiment (reference: Gaussian_experiment/) on the HIP ops: a self-contained UML regression fixture."""
