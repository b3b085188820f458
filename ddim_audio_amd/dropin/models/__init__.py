"""Drop-in for the reference's ``models`` directory (a namespace package upstream: no ``__init__.py``)."""
