"""Block task accounting of the drivers: retries, failed / orphaned counts and the error the reference raises.

Behavioural mirror of /root/reference/bootstrapper/blockwise.py:12-74 (`check_task_states`, `run_blockwise`) for
workers that walk their own block list instead of asking a daisy server for blocks: a block whose process function
raises is tried again up to `max_retries` more times (daisy's `max_retries`, predict.py:38: 5), then counted as failed;
a block of a downstream task whose upstream block failed is not run and counted as orphaned; any failed or orphaned
block ends the run with `RuntimeError("task <id>: <f> failed, <o> orphaned of <n> blocks")`.
"""

MAX_RETRIES = 5  # reference predict.py:38


class FatalBlockError(Exception):
    """An error after which the worker cannot go on (a HIP runtime error leaves the device context unusable): not
    retried; the worker process ends with it, like a crashed daisy worker (CalledProcessError in the reference)."""


class TaskState:
    def __init__(self, task_id, total_block_count=0):
        self.task_id = task_id
        self.total_block_count = int(total_block_count)
        self.completed_count = 0
        self.failed_count = 0
        self.orphaned_count = 0
        self.failed_blocks = []

    def merge(self, other):
        self.total_block_count += other.total_block_count
        self.completed_count += other.completed_count
        self.failed_count += other.failed_count
        self.orphaned_count += other.orphaned_count
        self.failed_blocks += other.failed_blocks
        return self

    def as_tuple(self):
        return (self.total_block_count, self.completed_count, self.failed_count, self.orphaned_count, list(self.failed_blocks))

    @classmethod
    def from_tuple(cls, task_id, t):
        s = cls(task_id, t[0])
        s.completed_count, s.failed_count, s.orphaned_count, s.failed_blocks = t[1], t[2], t[3], list(t[4])
        return s


def check_task_states(task_states):
    """blockwise.py:12-22"""
    errors = [f"task {task_id}: {ts.failed_count} failed, {ts.orphaned_count} orphaned of {ts.total_block_count} blocks"
              for task_id, ts in task_states.items() if ts.failed_count > 0 or ts.orphaned_count > 0]
    if errors:
        raise RuntimeError("; ".join(errors))


def is_fatal(exc):
    from ._lib import BsmiError, ERR_HIP
    return isinstance(exc, FatalBlockError) or (isinstance(exc, BsmiError) and exc.code == ERR_HIP)


def run_blocks(task_id, blocks, process_block, max_retries=MAX_RETRIES, upstream_failed=(), depends_on=None):
    """Run process_block(block) for every block of this worker.  upstream_failed: blocks of the upstream task that
    failed; depends_on(block) -> the upstream blocks it reads (a block with a failed upstream block is orphaned).
    -> TaskState (check_task_states raises on failures; a worker returns it to the parent, which merges)."""
    state = TaskState(task_id, len(blocks))
    bad = set(upstream_failed)
    for blk in blocks:
        if bad and depends_on is not None and any(u in bad for u in depends_on(blk)):
            state.orphaned_count += 1
            continue
        for attempt in range(max_retries + 1):
            try:
                process_block(blk)
                state.completed_count += 1
                break
            except Exception as exc:  # noqa: BLE001 - a block is retried like a daisy block
                if is_fatal(exc):
                    raise
                if attempt == max_retries:
                    state.failed_count += 1
                    state.failed_blocks.append(blk)
    return state
