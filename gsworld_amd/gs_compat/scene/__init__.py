"""``scene`` package of the 3DGS python layer, as far as GSWorld touches it: ``scene.cameras.Camera`` and
``scene.gaussian_model.GaussianModel``."""
