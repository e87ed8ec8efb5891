"""Drop-in entry point: same name and flags as the reference's ddsp_prematch_dataset.py (prematch / training-pool generation)."""
from knn_svc_amd.prematch import main

if __name__ == "__main__":
    raise SystemExit(main())
