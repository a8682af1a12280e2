from fadtk_amd.utils import *          # noqa: F401,F403
