"""`@hyrax_model` when Hyrax is installed, a no-op-compatible decorator otherwise.

The reference decorates its models with hyrax.models.hyrax_model (astrominn.py:67,
HyraxBaselineCLS.py:9, spectranet.py:85), which registers the class and lets Hyrax inject
`self.optimizer` / `self.criterion` from the config.  Hyrax is not a dependency of this path.
"""

try:  # pragma: no cover - hyrax is absent in the build image
    from hyrax.models import hyrax_model  # type: ignore
except Exception:  # ModuleNotFoundError or a partial install

    def hyrax_model(cls):
        return cls
