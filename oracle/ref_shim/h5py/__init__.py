"""Test-only empty stand-in (the reference imports h5py at module scope; nothing on the path uses it)."""
