import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
