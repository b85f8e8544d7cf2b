"""Exception hierarchy of the hot path, same names and argument meaning as the
reference (src/exceptions/montecarlo_exceptions.py:24-131,
src/exceptions/greek_exceptions.py:4-15)."""

__all__ = ["MonteCarloError", "InputValidationError", "ConvergenceError", "AccelerationError", "GreeksError"]


class MonteCarloError(Exception):
    def __init__(self, message: str = "Monte Carlo computation error"):
        self.message = message
        super().__init__(message)


class InputValidationError(MonteCarloError):
    def __init__(self, message: str = "Invalid input parameters"):
        super().__init__(f"Input validation failed: {message}")


class ConvergenceError(MonteCarloError):
    def __init__(self, message: str = "Simulation did not converge", iterations: int = 0):
        self.iterations = iterations
        super().__init__(f"{message} (after {iterations} iterations)")


class AccelerationError(MonteCarloError):
    """Raised for every libolmc / HIP failure (``backend="hip"``)."""

    def __init__(self, message: str = "Hardware acceleration failed", backend: str = "unknown"):
        self.backend = backend
        super().__init__(f"{message} (backend: {backend})")


class GreeksError(Exception):
    def __init__(self, message: str = "An error occurred in Greeks calculations."):
        super().__init__(message)
