"""Drop-in loader module: same public names as the reference's ddsp_hubconf.py (torch.hub looks for hubconf.py)."""
from knn_svc_amd.hubconf import dependencies, hifigan_wavlm, knn_vc, wavlm_large  # noqa: F401
