"""Test-only empty stand-in (the reference imports cvxpy at module scope; nothing on the path uses it)."""
