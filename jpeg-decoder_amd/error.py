"""Mirror of src/error.rs:16-48: Error::{Format, Unsupported, Io, Internal}."""
from . import _native as N


class Error(Exception):
    """Base of every error this package raises (``jpeg_decoder::Error``)."""
    kind = "Error"


class FormatError(Error):
    kind = "Format"


class UnsupportedError(Error):
    kind = "Unsupported"


class IoError(Error):
    kind = "Io"


class InternalError(Error):
    kind = "Internal"


class NoDeviceError(IoError):
    """No usable MI355X: there is no CPU fallback in the product path."""
    kind = "Io"


_BY_STATUS = {N.ERR_FORMAT: FormatError, N.ERR_UNSUPPORTED: UnsupportedError, N.ERR_IO: IoError,
              N.ERR_INTERNAL: InternalError, N.ERR_NO_DEVICE: NoDeviceError}


def check(status, message=b""):
    if status == N.OK:
        return
    if isinstance(message, bytes):
        message = message.decode(errors="replace")
    raise _BY_STATUS.get(status, Error)(message or N.lib().jpgpu_status_string(status).decode())


def error_for_status(status, message=b""):
    """The exception `check` would raise, as a value (per-image results of a pipeline)."""
    if isinstance(message, bytes):
        message = message.decode(errors="replace")
    return _BY_STATUS.get(status, Error)(message or N.lib().jpgpu_status_string(status).decode())
