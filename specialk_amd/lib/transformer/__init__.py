from . import Beam, Layers, Models, Modules, Optim, SubLayers, Translator  # noqa: F401

__all__ = ["Beam", "Layers", "Models", "Modules", "Optim", "SubLayers", "Translator"]
