"""Diagnostic scripts that replay a step / an export in the CPU oracle next to the HIP path (test infrastructure: they
import oracle/; run by hand on a GPU box, not collected by pytest)."""
