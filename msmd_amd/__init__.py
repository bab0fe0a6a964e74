"""Import alias for the package directory ``ubisoft-laforge-msmd_amd/``.

The directory name required by the repo layout contains hyphens and therefore
cannot be imported with a plain ``import`` statement.  This tiny package makes
``import msmd_amd.<module>`` resolve to ``ubisoft-laforge-msmd_amd/<module>.py``
by pointing its ``__path__`` at that directory.
"""
import os as _os

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
PKG_DIR = _os.path.join(_ROOT, "ubisoft-laforge-msmd_amd")
__path__.insert(0, PKG_DIR)

from msmd_amd._version import __version__  # noqa: E402,F401
