"""Test infrastructure only (see oracle/fluid_oracle.c header).

Nothing under fluidsimulationcuda_amd/ may import this package."""
