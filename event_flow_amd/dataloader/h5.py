"""Placeholder for the reference's HDF5 event reader (dataloader/h5.py): outside the hot path and not buildable in
this image (no h5py, no datasets).  The drivers accept `--synthetic` and use dataloader/synthetic_loader.py, which
honours the same batch contract; binding a real reader only needs to hand [B,N,4] event lists to
`dataloader.encodings.encode_event_list`."""


class H5Loader:
    def __init__(self, *args, **kwargs):
        raise ImportError(
            "H5Loader needs h5py and the DSEC/MVSEC/UZH-FPV HDF5 files; neither is available in this environment. "
            "Run the drivers with --synthetic (event_flow_amd.dataloader.synthetic_loader.SyntheticLoader)."
        )
