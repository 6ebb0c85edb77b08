"""Exception types (klang/errors.py:1-8 of the reference keeps a single base class)."""


class KlangError(Exception):

    """Base exception of klang_amd."""
