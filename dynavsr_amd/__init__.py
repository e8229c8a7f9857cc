"""MI355X-native EDVR hot path of DynaVSR.  Importing the package changes no process state."""


def configure_runtime(hw_queues=6):
    """Explicit, idempotent runtime set-up (hardware queues for the plans' side streams): call before the first HIP
    call of the process.  `models.create_model` and `bench.py` do.  See dynavsr_amd/_lib.py:configure_runtime."""
    from ._lib import configure_runtime as _c
    return _c(hw_queues)
