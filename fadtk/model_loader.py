from fadtk_amd.model_loader import *          # noqa: F401,F403
from fadtk_amd.model_loader import ModelLoader, get_all_models   # noqa: F401
