"""Drop-in entry point: same name and flags as the reference's ddsp_inference.py."""
from knn_svc_amd.inference import main

if __name__ == "__main__":
    raise SystemExit(main())
