from .mppi_delay import MPPIDelay  # noqa: F401
