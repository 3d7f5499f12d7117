"""Socket power and shader clock of one GPU from the amdgpu driver's hwmon files, sampled on a host thread while a phase runs
(measurement tooling: bench.py's energy leg, tools/layer_flood.py --power).  No rocm-smi process, no library: two sysfs reads per sample."""
import glob
import threading
import time


class Sampler:
    def __init__(self, device_index=0):
        self.dir = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            d = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            self.dir = d[0] if d else None
        except Exception as e:  # noqa: BLE001
            self.err = str(e)

    def available(self):
        return self.dir is not None

    def run(self, fn, period=0.005):
        """fn() with the sampler running; returns (fn's result, mean W, mean MHz, samples) over the SECOND half of the samples (the
        sensor averages over a window); (result, None, None, 0) when the files are not there"""
        if not self.dir:
            return fn(), None, None, 0
        stop = [False]
        pw, ck = [], []

        def loop():
            while not stop[0]:
                try:
                    pw.append(int(open(self.dir + "/power1_input").read()) / 1e6)
                    ck.append(int(open(self.dir + "/freq1_input").read()) / 1e6)
                except Exception:  # noqa: BLE001
                    pass
                time.sleep(period)
        t = threading.Thread(target=loop)
        t.start()
        r = fn()
        stop[0] = True
        t.join()
        half = len(pw) // 2
        if not pw[half:]:
            return r, None, None, 0
        return r, sum(pw[half:]) / len(pw[half:]), sum(ck[half:]) / max(len(ck[half:]), 1), len(pw[half:])
