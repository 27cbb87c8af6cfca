"""Drop-in module with the reference's name.  `import literate_library` / `from literate_library
import *` resolve to literate_amd.literate_library (the MI355X-backed function surface), aliased
rather than copied so that its module globals (`n_bins`, `sp_events_bin`, ...) stay live."""
import sys

from literate_amd import literate_library as _impl

sys.modules[__name__] = _impl
