from .noatt import MutanNoAtt  # noqa: F401
from .utils import factory, model_names  # noqa: F401
