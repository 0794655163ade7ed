"""Drop-in stand-in for the third-party ``nvdiffrast`` package, backed by liba3d_hip.so (see .torch)."""
__version__ = "0.3.1+a3d"
