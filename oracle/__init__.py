"""ORACLE -- test infrastructure only (see oracle/torch_ref.py). Never imported by the product package."""
