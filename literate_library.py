"""Drop-in module with the reference's name: `from literate_library import *` gives the
MI355X-backed function surface (see literate_amd/literate_library.py)."""
from literate_amd.literate_library import *  # noqa: F401,F403
from literate_amd.literate_library import __all__  # noqa: F401
